"""Import the REFERENCE's own python files in this container (TEST INFRASTRUCTURE).

``/root/reference/i2vgen-xl/pnp_utils.py`` and
``/root/reference/consisti2v/ddim_inverse_scheduler.py`` import a handful of symbols
from packages that are not installed here (``torchvision``, ``diffusers``).  This
module puts minimal stand-ins for exactly those symbols into ``sys.modules`` and then
imports the reference files *unmodified* from where they lie.  Used only to pin the
oracle / generate ``tests/golden`` fixtures; ``/root/reference`` does not exist on the
GPU box, so nothing on the ``-m gpu`` path calls this.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("ANYV2V_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "i2vgen-xl", "pnp_utils.py"))


def _mod(name: str, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []  # behave like a package
        sys.modules[name] = m
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def install_stubs():
    from oracle import unet_oracle as uo

    _mod("torchvision")
    _mod("torchvision.transforms")
    _mod("torchvision.io", read_video=None, write_video=None)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].io = sys.modules["torchvision.io"]

    class ConfigMixin:
        config_name = "scheduler_config.json"

    def register_to_config(init):
        import functools
        import inspect

        @functools.wraps(init)
        def wrapper(self, *a, **kw):
            sig = inspect.signature(init)
            bound = sig.bind(self, *a, **kw)
            bound.apply_defaults()
            cfg = {k: v for k, v in bound.arguments.items() if k not in ("self", "kwargs")}
            self.config = types.SimpleNamespace(**cfg)
            init(self, *a, **kw)
        return wrapper

    class SchedulerMixin:
        pass

    class BaseOutput:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    def _baseoutput_dc(cls):
        return cls

    _mod("diffusers")
    _mod("diffusers.utils", USE_PEFT_BACKEND=True, BaseOutput=object, deprecate=lambda *a, **k: None)
    _mod("diffusers.models")
    _mod("diffusers.models.upsampling", Upsample2D=uo.Upsample2D)
    _mod("diffusers.models.downsampling", Downsample2D=uo.Downsample2D)
    _mod("diffusers.models.attention_processor", AttnProcessor2_0=uo.AttnProcessor2_0)
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    _mod("diffusers.schedulers")
    _mod("diffusers.schedulers.scheduling_utils", SchedulerMixin=SchedulerMixin,
         KarrasDiffusionSchedulers=[])
    _mod("diffusers.utils.torch_utils", randn_tensor=None)


def _load(path: str, name: str):
    """Import one reference file with the stand-ins visible, then take them out of ``sys.modules`` again: the loaded
    module keeps the names it imported, and nothing else in the process (``transformers`` probes ``torchvision`` with
    ``importlib.util.find_spec``) ever sees a fake package."""
    before = set(sys.modules)
    install_stubs()
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k in set(sys.modules) - before:
            if k == "torchvision" or k.startswith("torchvision.") or k == "diffusers" or k.startswith("diffusers."):
                del sys.modules[k]
    return mod


def load_reference_pnp_utils():
    """The reference's ``i2vgen-xl/pnp_utils.py``, verbatim."""
    return _load(os.path.join(REFERENCE_ROOT, "i2vgen-xl", "pnp_utils.py"), "_ref_pnp_utils")


def load_reference_inverse_scheduler():
    """The reference's vendored ``consisti2v/ddim_inverse_scheduler.py``, verbatim."""
    return _load(os.path.join(REFERENCE_ROOT, "consisti2v", "ddim_inverse_scheduler.py"), "_ref_ddim_inverse")


def load_reference_consisti2v_models(attention_base=None, package="_ref_consisti2v_models", with_unet=False):
    """The reference's in-tree restatements of the diffusers building blocks, verbatim:
    ``consisti2v/consisti2v/models/videoldm_attention.py`` (``ConditionalAttention`` -- diffusers' ``Attention`` constructor,
    head reshapes and score arithmetic), ``videoldm_transformer_blocks.py`` (``BasicConditionalTransformerBlock`` /
    ``Transformer2DConditionModel`` -- block order, ``double_self_attention``, the spatial transformer's norm / proj / permute
    wrapper) and ``videoldm_unet_blocks.py`` (``Conv3DLayer``, the (3,1,1) temporal convolution).  Everything they import
    from diffusers is a stand-in: plain torch layers, and ``FeedForward`` is the ORACLE's own (its arithmetic lives in
    diffusers 0.26.3 and stays unpinned).  Returns (attention module, transformer-blocks module, unet-blocks module)."""
    import torch
    from torch import nn
    from oracle import unet_oracle as uo

    before = set(sys.modules)
    install_stubs()
    try:
        class _Logger:
            def __getattr__(self, k):
                return lambda *a, **kw: None

        class _Dummy(nn.Module):
            def __init__(self, *a, **kw):
                super().__init__()

        class LoRACompatibleLinear(nn.Linear):
            def forward(self, x, scale: float = 1.0):
                return super().forward(x)

        class LoRACompatibleConv(nn.Conv2d):
            def forward(self, x, scale: float = 1.0):
                return super().forward(x)

        class FeedForward(uo.FeedForward):  # diffusers signature in front of the oracle's arithmetic (NOT a pin of GEGLU)
            def __init__(self, dim, dropout=0.0, activation_fn="geglu", final_dropout=False, **kw):
                super().__init__(dim, None, activation_fn)

            def forward(self, x, scale: float = 1.0):
                return super().forward(x)

        class ResnetBlock2D(uo.ResnetBlock2D):  # diffusers signature in front of the oracle's block (pinned elsewhere: its body ==
            # i2vgen-xl/pnp_utils.py:46-126 and == seine/models/resnet.py:113-206 at one frame); LoRA-compatible layers because
            # consisti2v/pnp_utils.py:40-128 calls conv1(h, scale) / time_emb_proj(temb, scale)
            def __init__(self, *, in_channels, out_channels, temb_channels, eps=1e-5, groups=32, dropout=0.0,
                         time_embedding_norm="default", non_linearity="swish", output_scale_factor=1.0, pre_norm=True, **kw):
                assert time_embedding_norm == "default" and dropout == 0.0
                super().__init__(in_channels, out_channels, temb_channels, groups, eps)
                self.conv1 = LoRACompatibleConv(in_channels, out_channels, 3, padding=1)
                self.conv2 = LoRACompatibleConv(out_channels, out_channels, 3, padding=1)
                self.time_emb_proj = LoRACompatibleLinear(temb_channels, out_channels)
                if in_channels != out_channels:
                    self.conv_shortcut = LoRACompatibleConv(in_channels, out_channels, 1)
                self.output_scale_factor = output_scale_factor

        class Upsample2D(uo.Upsample2D):
            def __init__(self, channels, use_conv=True, out_channels=None, **kw):
                assert use_conv and (out_channels is None or out_channels == channels)
                super().__init__(channels)

            def forward(self, x, output_size=None, scale=1.0):
                return super().forward(x, output_size)      # ([3P] nearest x 2, or nearest to ``output_size``: oracle/unet_oracle.py)

        class Transformer2DModelOutput:
            def __init__(self, sample):
                self.sample = sample

        ident = lambda cls: cls
        _mod("diffusers.utils", USE_PEFT_BACKEND=True, BaseOutput=object, deprecate=lambda *a, **k: None,
             logging=types.SimpleNamespace(get_logger=lambda *a, **k: _Logger()), is_torch_version=lambda *a, **k: True)
        _mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
        _mod("diffusers.utils.torch_utils", randn_tensor=None, maybe_allow_in_graph=ident)
        _mod("diffusers.models.lora", LoRACompatibleLinear=LoRACompatibleLinear, LoRACompatibleConv=LoRACompatibleConv,
             LoRALinearLayer=_Dummy)
        # ``TemporalConditionalAttention`` subclasses diffusers' ``Attention``; load_reference_consisti2v_decoder() passes the
        # reference's own in-tree ``ConditionalAttention`` (the same constructor, restated by the reference) as that base
        ap = _mod("diffusers.models.attention_processor", AttnProcessor2_0=uo.AttnProcessor2_0, Attention=attention_base or _Dummy)
        for n in ("AttnAddedKVProcessor", "AttnAddedKVProcessor2_0", "AttnProcessor", "SpatialNorm", "CustomDiffusionAttnProcessor",
                  "CustomDiffusionXFormersAttnProcessor", "SlicedAttnAddedKVProcessor", "XFormersAttnAddedKVProcessor",
                  "LoRAAttnAddedKVProcessor", "XFormersAttnProcessor", "LoRAXFormersAttnProcessor", "LoRAAttnProcessor",
                  "LoRAAttnProcessor2_0", "SlicedAttnProcessor", "AttentionProcessor"):
            setattr(ap, n, type(n, (), {}))
        ap.LORA_ATTENTION_PROCESSORS = ()
        _mod("diffusers.models.embeddings", ImagePositionalEmbeddings=_Dummy, PatchEmbed=_Dummy)
        _mod("diffusers.models.attention", AdaLayerNorm=_Dummy, AdaLayerNormZero=_Dummy, FeedForward=FeedForward,
             GatedSelfAttentionDense=_Dummy)
        _mod("diffusers.models.modeling_utils", ModelMixin=nn.Module)
        _mod("diffusers.models.transformer_2d", Transformer2DModelOutput=Transformer2DModelOutput)
        class Downsample2D(uo.Downsample2D):   # diffusers signature (use_conv=True, stride-2 3x3 conv, padding 1, name "op" -> key `conv`)
            def __init__(self, channels, use_conv=True, out_channels=None, padding=1, name="conv", **kw):
                assert use_conv and padding == 1 and (out_channels is None or out_channels == channels)
                super().__init__(channels)

            def forward(self, x, scale=1.0):
                return super().forward(x)

        class DownBlock2D(nn.Module):   # diffusers DownBlock2D as VideoLDMDownBlock uses it: positional constructor, resnets + downsamplers
            def __init__(self, in_channels, out_channels, temb_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6,
                         resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32, resnet_pre_norm=True,
                         output_scale_factor=1.0, add_downsample=True, downsample_padding=1):
                super().__init__()
                self.resnets = nn.ModuleList([
                    ResnetBlock2D(in_channels=in_channels if i == 0 else out_channels, out_channels=out_channels, temb_channels=temb_channels,
                                  eps=resnet_eps, groups=resnet_groups, dropout=dropout, time_embedding_norm=resnet_time_scale_shift,
                                  non_linearity=resnet_act_fn, output_scale_factor=output_scale_factor, pre_norm=resnet_pre_norm)
                    for i in range(num_layers)])
                self.downsamplers = nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                                padding=downsample_padding, name="op")]) if add_downsample else None
                self.gradient_checkpointing = False

        class UpBlock2D(nn.Module):
            def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6,
                         resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32, resnet_pre_norm=True,
                         output_scale_factor=1.0, add_upsample=True):
                super().__init__()
                self.resnets = nn.ModuleList()
                for i in range(num_layers):
                    res_skip = in_channels if i == num_layers - 1 else out_channels
                    res_in = prev_output_channel if i == 0 else out_channels
                    self.resnets.append(ResnetBlock2D(in_channels=res_in + res_skip, out_channels=out_channels, temb_channels=temb_channels,
                                                      eps=resnet_eps, groups=resnet_groups, dropout=dropout,
                                                      time_embedding_norm=resnet_time_scale_shift, non_linearity=resnet_act_fn,
                                                      output_scale_factor=output_scale_factor, pre_norm=resnet_pre_norm))
                self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)]) if add_upsample else None
                self.gradient_checkpointing = False

        _mod("diffusers.models.unet_2d_blocks", DownBlock2D=DownBlock2D, UpBlock2D=UpBlock2D, UNetMidBlock2DCrossAttn=_Dummy,
             UNetMidBlock2DSimpleCrossAttn=_Dummy)
        _mod("diffusers.models.resnet", ResnetBlock2D=ResnetBlock2D, Downsample2D=Downsample2D, Upsample2D=Upsample2D)
        _mod("diffusers.models.dual_transformer_2d", DualTransformer2DModel=_Dummy)
        _mod("diffusers.models.activations", get_activation=lambda name: nn.SiLU())
        import typing
        _mod("beartype", beartype=ident)
        _mod("beartype.typing", Literal=typing.Literal, Union=typing.Union, Optional=typing.Optional)
        pkg_dir = os.path.join(REFERENCE_ROOT, "consisti2v", "consisti2v", "models")
        pkg = types.ModuleType(package)
        pkg.__path__ = [pkg_dir]
        sys.modules[package] = pkg
        import importlib
        att = importlib.import_module(package + ".videoldm_attention")
        blocks = importlib.import_module(package + ".videoldm_transformer_blocks")
        ublocks = importlib.import_module(package + ".videoldm_unet_blocks")
        if with_unet:
            _install_unet_stubs(nn, _Dummy, _Logger)
            ublocks.unet = importlib.import_module(package + ".videoldm_unet")
    finally:
        for k in set(sys.modules) - before:
            if k.split(".")[0] in ("torchvision", "diffusers", "beartype"):
                del sys.modules[k]
    return att, blocks, ublocks


def _install_unet_stubs(nn, _Dummy, _Logger):
    """What ``consisti2v/consisti2v/models/videoldm_unet.py:1-62`` imports from diffusers beyond the block files' needs.  The two
    embedding classes are restated from the published diffusers-0.21 code (``Timesteps`` = sinusoidal embedding with
    ``flip_sin_to_cos`` / ``downscale_freq_shift``; ``TimestepEmbedding`` = Linear -> act -> Linear); everything else is only
    referenced by name."""
    import math
    import torch

    class Timesteps(nn.Module):
        def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
            super().__init__()
            self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift

        def forward(self, timesteps):
            half = self.num_channels // 2
            exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / (half - self.downscale_freq_shift)
            emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
            emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
            if self.flip_sin_to_cos:
                emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
            return emb

    class TimestepEmbedding(nn.Module):
        def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None):
            super().__init__()
            assert post_act_fn is None and cond_proj_dim is None and act_fn in ("silu", "swish")
            self.linear_1 = nn.Linear(in_channels, time_embed_dim)
            self.act = nn.SiLU()
            self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

        def forward(self, sample, condition=None):
            return self.linear_2(self.act(self.linear_1(sample)))

    class UNet2DConditionOutput:
        def __init__(self, sample):
            self.sample = sample

    emb = _mod("diffusers.models.embeddings", ImagePositionalEmbeddings=_Dummy, PatchEmbed=_Dummy, Timesteps=Timesteps,
               TimestepEmbedding=TimestepEmbedding)
    for n in ("GaussianFourierProjection", "ImageHintTimeEmbedding", "ImageProjection", "ImageTimeEmbedding", "PositionNet",
              "TextImageProjection", "TextImageTimeEmbedding", "TextTimeEmbedding"):
        setattr(emb, n, _Dummy)
    ap = sys.modules["diffusers.models.attention_processor"]
    ap.ADDED_KV_ATTENTION_PROCESSORS, ap.CROSS_ATTENTION_PROCESSORS = (), ()
    _mod("diffusers.loaders", UNet2DConditionLoadersMixin=type("UNet2DConditionLoadersMixin", (), {}))
    sys.modules["diffusers.models"].ModelMixin = nn.Module
    _mod("diffusers.models.unet_2d_condition", UNet2DConditionOutput=UNet2DConditionOutput)
    mu = sys.modules["diffusers.models.modeling_utils"]
    mu.load_state_dict = mu.load_model_dict_into_meta = None
    du = sys.modules["diffusers.utils"]
    for n in ("CONFIG_NAME", "DIFFUSERS_CACHE", "FLAX_WEIGHTS_NAME", "HF_HUB_OFFLINE", "SAFETENSORS_WEIGHTS_NAME", "WEIGHTS_NAME",
              "_add_variant", "_get_model_file"):
        setattr(du, n, None)
    du.is_accelerate_available = lambda: False
    sys.modules["diffusers"].__version__ = "0.21.2"


def load_reference_consisti2v_unet():
    """The reference's WHOLE ``VideoLDMUNet3DConditionModel`` (``consisti2v/consisti2v/models/videoldm_unet.py:68-1064``: conv_in, time +
    frame-stride embeddings, the four encoder blocks, the mid block, the four decoder blocks, conv_out, the first-frame "concat"
    conditioning of ``forward``), imported verbatim on top of the block files of ``load_reference_consisti2v_decoder`` -- with the
    reference's own in-tree ``ConditionalAttention`` as the base class of ``TemporalConditionalAttention`` -- and the hook functions
    of ``consisti2v/pnp_utils.py``.  Returns (unet module, unet-blocks module, pnp_utils module)."""
    att1, _, _ = load_reference_consisti2v_models()
    att, blocks, ublocks = load_reference_consisti2v_models(attention_base=att1.ConditionalAttention,
                                                            package="_ref_consisti2v_models_full", with_unet=True)
    from oracle import unet_oracle as uo
    before = set(sys.modules)
    install_stubs()
    try:
        _mod("diffusers.models.resnet", Upsample2D=uo.Upsample2D, Downsample2D=uo.Downsample2D)
        _mod("diffusers.models.attention_processor", AttnProcessor2_0=uo.AttnProcessor2_0, Attention=att.ConditionalAttention)
        _mod("consisti2v")
        _mod("consisti2v.models")
        sys.modules["consisti2v.models.videoldm_attention"] = att
        spec = importlib.util.spec_from_file_location("_ref_consisti2v_pnp_utils_full",
                                                      os.path.join(REFERENCE_ROOT, "consisti2v", "pnp_utils.py"))
        pnp = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(pnp)
    finally:
        for k in set(sys.modules) - before:
            if k.split(".")[0] in ("torchvision", "diffusers", "consisti2v"):
                del sys.modules[k]
    return ublocks.unet, ublocks, pnp


def load_reference_consisti2v_decoder():
    """Everything the ConsistI2V hook family touches, from the reference's own files: ``VideoLDMCrossAttnUpBlock``
    (``consisti2v/consisti2v/models/videoldm_unet_blocks.py:548-745`` -- ResnetBlock2D / TemporalResnetBlock / spatial and temporal
    ``Transformer2DConditionModel`` per layer), ``TemporalConditionalAttention`` with the rotary embedding
    (``videoldm_attention.py:552-641``, ``rotary_embedding.py``) and the hook functions of ``consisti2v/pnp_utils.py:19-345``.
    ``TemporalConditionalAttention`` derives from diffusers' ``Attention`` (not installed): the module is imported a second time
    with the reference's in-tree ``ConditionalAttention`` of the first import as that base class (the reference's own copy of the
    same constructor and helpers).  Returns (attention module, transformer-blocks module, unet-blocks module, pnp_utils module)."""
    att1, _, _ = load_reference_consisti2v_models()
    att, blocks, ublocks = load_reference_consisti2v_models(attention_base=att1.ConditionalAttention,
                                                            package="_ref_consisti2v_models_dec")
    from oracle import unet_oracle as uo
    before = set(sys.modules)
    install_stubs()
    try:
        _mod("diffusers.models.resnet", Upsample2D=uo.Upsample2D, Downsample2D=uo.Downsample2D)
        _mod("diffusers.models.attention_processor", AttnProcessor2_0=uo.AttnProcessor2_0, Attention=att.ConditionalAttention)
        _mod("consisti2v")
        _mod("consisti2v.models")
        sys.modules["consisti2v.models.videoldm_attention"] = att
        spec = importlib.util.spec_from_file_location("_ref_consisti2v_pnp_utils",
                                                      os.path.join(REFERENCE_ROOT, "consisti2v", "pnp_utils.py"))
        pnp = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(pnp)
    finally:
        for k in set(sys.modules) - before:
            if k.split(".")[0] in ("torchvision", "diffusers", "consisti2v"):
                del sys.modules[k]
    return att, blocks, ublocks, pnp


def load_reference_seine_blocks():
    """``seine/models/resnet.py`` (``ResnetBlock3D`` / ``Upsample3D`` / ``Downsample3D``: diffusers' ResNet block and samplers with
    per-frame 2-D convolutions on [b, c, f, h, w]; at f = 1 they ARE ``ResnetBlock2D`` / ``Upsample2D`` / ``Downsample2D``) and
    ``seine/models/utils.py`` (``timestep_embedding``), verbatim.  Both files import only torch / numpy / einops."""
    res = _load(os.path.join(REFERENCE_ROOT, "seine", "models", "resnet.py"), "_ref_seine_resnet")
    utl = _load(os.path.join(REFERENCE_ROOT, "seine", "models", "utils.py"), "_ref_seine_utils")
    return res, utl


def load_reference_seine_decoder(with_unet=False):
    """Everything SEINE's hook family touches, from the reference's own files: ``CrossAttnUpBlock3D``
    (``seine/models/unet_blocks.py:444-575``: ``ResnetBlock3D`` + ``Transformer3DModel`` per layer, ``Upsample3D``), the attention
    classes of ``seine/models/attention.py`` (``CrossAttention``, ``TemporalAttention`` with ``RelativePositionBias`` and the rotary
    embedding, ``BasicTransformerBlock``, ``Transformer3DModel``) and the hook functions of ``seine/pnp_utils.py:121-458``.
    ``rotary_embedding_torch`` (not installed) is the library the reference vendors as
    ``consisti2v/consisti2v/models/rotary_embedding.py`` -- that file is imported in its place; diffusers' ``FeedForward`` is the
    oracle's (unpinned, as everywhere).  Returns (attention module, unet_blocks module, resnet module, pnp_utils module,
    RotaryEmbedding class).  ``with_unet``: also the WHOLE ``UNet3DConditionModel`` (``seine/models/unet.py:98-560``, imported verbatim),
    as ``unet_blocks module.unet``."""
    import importlib
    import torch
    from torch import nn
    from oracle import unet_oracle as uo
    att_c2, _, _ = load_reference_consisti2v_models()
    rotary_mod = sys.modules[att_c2.__name__.rsplit(".", 1)[0] + ".rotary_embedding"]
    before = set(sys.modules)
    install_stubs()
    try:
        class FeedForward(uo.FeedForward):
            def __init__(self, dim, dropout=0.0, activation_fn="geglu", **kw):
                super().__init__(dim, None, activation_fn)

        class _Dummy(nn.Module):
            def __init__(self, *a, **kw):
                super().__init__()

        _mod("diffusers.utils", BaseOutput=object, deprecate=lambda *a, **k: None)
        _mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
        _mod("diffusers.models.attention", FeedForward=FeedForward, AdaLayerNorm=_Dummy)
        _mod("diffusers.models.modeling_utils", ModelMixin=nn.Module)
        _mod("rotary_embedding_torch", RotaryEmbedding=rotary_mod.RotaryEmbedding)
        pkg = types.ModuleType("_ref_seine_models")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "seine", "models")]
        sys.modules["_ref_seine_models"] = pkg
        path0 = list(sys.path)
        att = importlib.import_module("_ref_seine_models.attention")
        res = importlib.import_module("_ref_seine_models.resnet")
        ublocks = importlib.import_module("_ref_seine_models.unet_blocks")
        if with_unet:
            # ``seine/models/unet.py:16-23``: the config decorator, the embedding classes (restated as for ConsistI2V) and names only
            import inspect

            class _Logger:
                def __getattr__(self, k):
                    return lambda *a, **kw: None

            def register_to_config(init):
                sig = inspect.signature(init)

                def wrapped(self, *a, **kw):
                    bound = sig.bind(self, *a, **kw)
                    bound.apply_defaults()
                    object.__setattr__(self, "_cfg", types.SimpleNamespace(**{k: v for k, v in bound.arguments.items() if k != "self"}))
                    init(self, *a, **kw)
                return wrapped

            class ConfigMixin:
                @property
                def config(self):
                    return self._cfg

            class ModelMixin(nn.Module):
                @property
                def dtype(self):
                    return next(self.parameters()).dtype
            _install_unet_stubs(nn, _Dummy, _Logger)
            _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
            _mod("diffusers.utils", BaseOutput=object, deprecate=lambda *a, **k: None,
                 logging=types.SimpleNamespace(get_logger=lambda *a, **k: _Logger()))
            _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
            ublocks.unet = importlib.import_module("_ref_seine_models.unet")
        spec = importlib.util.spec_from_file_location("_ref_seine_pnp_utils", os.path.join(REFERENCE_ROOT, "seine", "pnp_utils.py"))
        pnp = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(pnp)
        sys.path[:] = path0   # (the reference files append their parent directory to sys.path)
    finally:
        for k in set(sys.modules) - before:
            if k.split(".")[0] in ("torchvision", "diffusers", "rotary_embedding_torch"):
                del sys.modules[k]
    return att, ublocks, res, pnp, rotary_mod.RotaryEmbedding
