"""Run the REFERENCE's own pipeline class on the CPU (TEST INFRASTRUCTURE; needs /root/reference, this container only).

``/root/reference/i2vgen-xl/pipelines/pipeline_i2vgen_xl.py`` is imported verbatim (``load_reference_pipeline_module``): its
``I2VGenXLPipeline.invert`` (``:1197-1451``), ``sample_with_pnp`` (``:892-1193``) and ``__call__`` (``:652-888``) loops, its
``encode_prompt`` / ``_encode_image`` / ``prepare_image_latents`` / ``encode_vae_video`` glue and its sibling modules
``pnp_utils.py`` / ``utils.py`` then run unmodified.  What the file imports from diffusers / torchvision is replaced by small
stand-ins (a ``DiffusionPipeline`` base with ``register_modules`` / ``progress_bar`` / ``_execution_device``; a
``VaeImageProcessor`` with the two conversions the pipeline calls), and the seven pipeline components are deterministic toys with
the interfaces the reference calls on them (``ToyTokenizer`` ... ``ToyVAE``), plus

* the UNet: the oracle ``I2VGenXLUNetOracle`` (any width),
* the inversion scheduler: the reference's own vendored ``consisti2v/ddim_inverse_scheduler.py``,
* the sampling scheduler: ``ForwardDDIM`` -- diffusers' ``DDIMScheduler`` is absent, so this adapter exposes its interface
  (``set_timesteps / timesteps / scale_model_input / step().prev_sample / init_noise_sigma / order``) over
  ``oracle.schedulers_oracle.ddim_step``.

Used by ``tests/test_oracle.py`` to pin ``oracle.pnp_oracle``'s loops and by ``tests/test_host_logic.py`` to compare the native
pipeline (op emulation) with the reference pipeline end to end.
"""
from __future__ import annotations

import contextlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from torch import nn

from . import ref_stubs
from . import schedulers_oracle as so


# ----------------------------------------------------------------------------------------------- module loading
def load_reference_pipeline_module():
    """(pipeline module, pnp_utils module, utils module) of the reference's ``i2vgen-xl/`` tree, verbatim."""
    import PIL.Image
    import transformers  # noqa: F401  (the pipeline file imports CLIP class names from the real package: load it before the stand-ins)

    before = set(sys.modules)
    ref_stubs.install_stubs()
    root = os.path.join(ref_stubs.REFERENCE_ROOT, "i2vgen-xl")

    class _Logger:
        def __getattr__(self, k):
            return lambda *a, **kw: None

    class BaseOutput:  # the reference's output classes are @dataclass subclasses of it
        pass

    class DiffusionPipeline:
        def __init__(self):
            self._device = torch.device("cpu")

        def register_modules(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        @property
        def device(self):
            return self._device

        @property
        def _execution_device(self):
            return self._device

        @contextlib.contextmanager
        def progress_bar(self, iterable=None, total=None):
            yield types.SimpleNamespace(update=lambda *a, **k: None)

        def maybe_free_model_hooks(self):
            pass

    class VaeImageProcessor:
        """``pil_to_numpy`` / ``numpy_to_pt`` / ``preprocess`` (``do_resize=False``: PIL -> [-1, 1] NCHW) / ``postprocess``."""

        def __init__(self, vae_scale_factor=8, do_resize=False):
            assert not do_resize

        @staticmethod
        def pil_to_numpy(images):
            images = images if isinstance(images, list) else [images]
            return np.stack([np.array(i).astype(np.float32) / 255.0 for i in images], 0)

        @staticmethod
        def numpy_to_pt(images):
            return torch.from_numpy(images.transpose(0, 3, 1, 2))

        def preprocess(self, image):
            return 2.0 * self.numpy_to_pt(self.pil_to_numpy(image)) - 1.0

        @staticmethod
        def postprocess(x, output_type="pil"):
            """[3P] diffusers 0.26 ``VaeImageProcessor.postprocess``: denormalise to [0, 1]; "pt" -> the tensor, "np" -> float32
            [n, h, w, c] (``pt_to_numpy``), "pil" -> ``numpy_to_pil`` ((x * 255).round() as uint8)."""
            x = (x / 2 + 0.5).clamp(0, 1)
            if output_type == "pt":
                return x
            arr = x.cpu().permute(0, 2, 3, 1).float().numpy()
            if output_type == "np":
                return arr
            return [PIL.Image.fromarray(a) for a in (arr * 255).round().astype("uint8")]

    m = ref_stubs._mod
    m("diffusers", DiffusionPipeline=DiffusionPipeline)
    m("diffusers.image_processor", PipelineImageInput=object, VaeImageProcessor=VaeImageProcessor)
    m("diffusers.loaders", LoraLoaderMixin=type("LoraLoaderMixin", (), {}))
    m("diffusers.models", AutoencoderKL=object)
    m("diffusers.models.lora", adjust_lora_scale_text_encoder=lambda *a, **k: None)
    m("diffusers.models.unets")
    m("diffusers.models.unets.unet_i2vgen_xl", I2VGenXLUNet=object)
    m("diffusers.schedulers", DDIMScheduler=object)
    m("diffusers.utils", USE_PEFT_BACKEND=True, BaseOutput=BaseOutput, deprecate=lambda *a, **k: None,
      logging=types.SimpleNamespace(get_logger=lambda *a, **k: _Logger()),
      replace_example_docstring=lambda doc: (lambda f: f), scale_lora_layers=lambda *a, **k: None,
      unscale_lora_layers=lambda *a, **k: None, load_image=lambda p: PIL.Image.open(p).convert("RGB"))
    m("diffusers.utils.torch_utils", randn_tensor=lambda shape, generator=None, device=None, dtype=None:
      torch.randn(shape, generator=generator, dtype=dtype).to(device))
    m("torchvision.transforms")
    try:
        mods = {}
        for name in ("pnp_utils", "utils"):  # the pipeline file imports its siblings by these bare names
            spec = importlib.util.spec_from_file_location(name, os.path.join(root, f"{name}.py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
            mods[name] = mod
        spec = importlib.util.spec_from_file_location("_ref_pipeline_i2vgen_xl", os.path.join(root, "pipelines", "pipeline_i2vgen_xl.py"))
        pm = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(pm)
    finally:
        for k in set(sys.modules) - before:
            if k.split(".")[0] in ("torchvision", "diffusers", "pnp_utils", "utils"):
                del sys.modules[k]
    return pm, mods["pnp_utils"], mods["utils"]


# ----------------------------------------------------------------------------------------------- toy components
class ToyTokenizer:
    model_max_length = 77  # CLIP's: the UNet context is 77 text + 64 image-latent + 4 image-embedding tokens = 145

    def __call__(self, prompts, padding=None, max_length=None, truncation=None, return_tensors=None):
        prompts = [prompts] if isinstance(prompts, str) else list(prompts)
        n = max_length if padding == "max_length" else max([len(p) + 1 for p in prompts] + [1])
        ids = torch.zeros(len(prompts), n, dtype=torch.long)
        for i, p in enumerate(prompts):
            t = [(ord(c) % 90) + 3 for c in p][: n - 1]
            if t:
                ids[i, : len(t)] = torch.tensor(t, dtype=torch.long)
            ids[i, len(t)] = 2
        return types.SimpleNamespace(input_ids=ids, attention_mask=(ids != 0).long())

    def batch_decode(self, ids):
        return ["?"] * len(ids)


class _TextOut(tuple):
    """tuple-style AND attribute-style access, like a transformers ModelOutput (the reference indexes, adapters use attributes)."""

    @property
    def hidden_states(self):
        return self[-1]

    @property
    def last_hidden_state(self):
        return self[0]


class ToyTextEncoder(nn.Module):
    """``CLIPTextModel``'s calling convention: ``enc(ids)[0]`` = final-LayerNorm'ed last hidden state; with
    ``output_hidden_states=True`` ``enc(ids)[-1]`` = tuple of hidden states; ``enc.text_model.final_layer_norm``; ``enc.config``."""

    def __init__(self, dim, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.emb = nn.Embedding(100, dim)
        self.pos = nn.Parameter(torch.randn(77, dim, generator=g) * 0.1)
        self.l1, self.l2 = nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.text_model = types.SimpleNamespace(final_layer_norm=nn.LayerNorm(dim))
        self._ln = self.text_model.final_layer_norm
        self.config = types.SimpleNamespace()
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.1))

    @property
    def dtype(self):
        return self.emb.weight.dtype

    def forward(self, ids, attention_mask=None, output_hidden_states=False):
        h0 = self.emb(ids) + self.pos[: ids.shape[1]]
        h1 = h0 + torch.tanh(self.l1(h0))
        h2 = h1 + torch.tanh(self.l2(h1))
        last = self._ln(h2)
        return _TextOut((last, None, (h0, h1, h2))) if output_hidden_states else _TextOut((last,))


class ToyImageEncoder(nn.Module):
    def __init__(self, dim, seed=1):
        super().__init__()
        self.proj = nn.Linear(3 * 8 * 8, dim)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)

    def forward(self, image=None, pixel_values=None):
        image = image if image is not None else pixel_values
        x = torch.nn.functional.adaptive_avg_pool2d(image.float(), 8).flatten(1)
        return types.SimpleNamespace(image_embeds=self.proj(x).to(image.dtype))


class ToyFeatureExtractor:
    crop_size = {"width": 224, "height": 224}
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)

    def __call__(self, images, do_normalize=True, do_center_crop=False, do_resize=False, do_rescale=False, return_tensors="pt"):
        assert do_normalize and not (do_center_crop or do_resize or do_rescale)
        x = images if torch.is_tensor(images) else torch.as_tensor(np.asarray(images))
        return types.SimpleNamespace(pixel_values=(x - torch.tensor(self.mean).view(1, 3, 1, 1)) / torch.tensor(self.std).view(1, 3, 1, 1))


class ToyVAE(nn.Module):
    """8 x 8 average pooling + a fixed 3 -> 4 channel mix as the 'posterior mean'; ``sample()`` adds global-RNG noise like the
    reference's ``latent_dist.sample()`` (std 0 by default so that runs are comparable without replaying the RNG stream)."""

    def __init__(self, noise_std=0.0):
        super().__init__()
        self.config = types.SimpleNamespace(scaling_factor=0.18215, block_out_channels=(1, 1, 1, 1))
        self.mix = nn.Parameter(torch.tensor([[0.6, 0.3, 0.1], [-0.4, 0.5, 0.2], [0.2, -0.3, 0.7], [0.3, 0.3, -0.5]]))
        self.noise_std = noise_std

    def encode(self, x):
        mean = torch.einsum("oc,nchw->nohw", self.mix.to(x.dtype), torch.nn.functional.avg_pool2d(x, 8)) * 4.0
        std = self.noise_std
        return types.SimpleNamespace(latent_dist=types.SimpleNamespace(
            sample=lambda: mean + (torch.randn(mean.shape).to(mean) * std if std else 0.0), mode=lambda: mean))

    def decode(self, z):
        x = torch.einsum("oc,nohw->nchw", self.mix.to(z.dtype), z / 4.0)
        return types.SimpleNamespace(sample=torch.nn.functional.interpolate(x, scale_factor=8.0, mode="nearest"))


class ForwardDDIM:
    """diffusers ``DDIMScheduler`` interface (config of ``i2vgen-xl/demo.ipynb:1208-1226``: v-prediction, eta 0, trailing-free
    'leading' spacing with steps_offset 1, zero terminal SNR) over ``oracle.schedulers_oracle``."""
    order, init_noise_sigma = 1, 1.0

    def __init__(self):
        self.ac = so.alphas_cumprod()
        self.timesteps, self.n = None, None

    def set_timesteps(self, n, device=None):
        self.n = n
        self.timesteps = torch.from_numpy(so.ddim_timesteps(n).copy())

    def scale_model_input(self, x, t):
        return x

    def step(self, model_output, timestep, sample, eta=0.0, generator=None):
        assert eta == 0.0
        out = so.ddim_step(model_output.double().numpy(), int(timestep), sample.double().numpy(), self.n, self.ac)
        return types.SimpleNamespace(prev_sample=torch.from_numpy(np.asarray(out)).to(sample.dtype))


def build_reference_pipeline(unet_oracle, dim, vae_noise_std=0.0):
    """The reference's ``I2VGenXLPipeline`` (its real ``__init__``) around the oracle UNet and the toy components.  ``dim`` = the
    UNet's cross_attention_dim (text / image embedding width).  Returns (pipeline, pipeline module, pnp_utils module)."""
    pm, pnp, _ = load_reference_pipeline_module()
    pipe = pm.I2VGenXLPipeline(vae=ToyVAE(vae_noise_std), text_encoder=ToyTextEncoder(dim), tokenizer=ToyTokenizer(),
                               image_encoder=ToyImageEncoder(dim), feature_extractor=ToyFeatureExtractor(), unet=unet_oracle,
                               scheduler=ForwardDDIM())
    return pipe, pm, pnp


INV_CFG = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="squaredcos_cap_v2", clip_sample=False,
               set_alpha_to_one=True, steps_offset=1, prediction_type="v_prediction", timestep_spacing="leading",
               rescale_betas_zero_snr=True)  # i2vgen-xl/demo.ipynb:1208-1226


@torch.no_grad()
def run_reference_job(oracle_unet, dim, frames, edited, size, n_steps, ratios, work_dir, neg="blurry", edit_prompt="a robot",
                      with_reconstruction=True):
    """Stage 1 and stage 2 of the reference on ONE clip, driven exactly as its two runners drive the pipeline
    (``run_group_ddim_inversion.py:29-77``, ``run_group_pnp_edit.py:35-47,107-140``): ``encode_vae_video`` -> ``invert`` (cfg 1,
    empty prompt, fps 8, files written to ``work_dir``) -> [``__call__`` CFG reconstruction from the noisiest latent] -> the
    reference's ``register_*`` hooks with ``int(n_steps * ratio)``-long schedule prefixes -> ``sample_with_pnp`` (cfg 9, t_idx 0).
    Returns everything a comparison needs: the conditioning tensors the reference's glue code built, the trajectory it wrote, the
    edited / reconstructed latents.  ``oracle_unet`` is left with the reference hooks installed (pass a copy)."""
    import copy
    import warnings
    Fr = len(frames)
    dev = torch.device("cpu")
    plain = copy.deepcopy(oracle_unet) if with_reconstruction else None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref, pm, ref_pnp = build_reference_pipeline(oracle_unet, dim)
        inv_mod = ref_stubs.load_reference_inverse_scheduler()
    ref.scheduler = inv_mod.DDIMInverseScheduler(**INV_CFG)
    lat0 = ref.encode_vae_video(frames, device=dev, height=size, width=size)
    out_dir = os.path.join(str(work_dir), "ddim_latents")
    inverted = ref.invert(prompt="", image=frames[0], height=size, width=size, num_frames=Fr, num_inference_steps=n_steps,
                          guidance_scale=1.0, negative_prompt=neg, target_fps=8, latents=lat0,
                          generator=torch.Generator().manual_seed(8888), return_dict=False, output_dir=out_dir)
    inv_ts = [int(t) for t in ref.scheduler.timesteps]
    files = {t: torch.load(os.path.join(out_dir, f"ddim_latents_{t}.pt")) for t in inv_ts}
    ref._guidance_scale = 1.0
    src_pe, _ = ref.encode_prompt("", dev, 1, None, clip_skip=1)
    crop = lambda im: pm._resize_bilinear(pm._center_crop_wide(im, (size, size)), (224, 224))
    prep = lambda im: ref.image_processor.preprocess(pm._center_crop_wide(im, (size, size)))
    src_ie = ref._encode_image(crop(frames[0]), dev, 1)
    src_il = ref.prepare_image_latents(prep(frames[0]), device=dev, num_frames=Fr, num_videos_per_prompt=1)
    fwd = ForwardDDIM()
    fwd.set_timesteps(n_steps)
    ts = fwd.timesteps
    T = int(ts[0])
    out = dict(lat0=lat0, inv_ts=inv_ts, files=files, inverted=inverted, src_pe=src_pe, src_ie=src_ie, src_il=src_il, T=T,
               out_dir=out_dir, ratios=tuple(ratios), n_steps=n_steps, neg=neg)
    if with_reconstruction:  # the reference reconstructs before any hook exists (stage 1): a hook-free pipeline
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref2, _, _ = build_reference_pipeline(plain, dim)
        ref2.scheduler = ForwardDDIM()
        out["rec_ref"] = ref2(prompt="", image=frames[0], height=size, width=size, num_frames=Fr, num_inference_steps=n_steps,
                              guidance_scale=9.0, negative_prompt=neg, target_fps=8, latents=files[T].clone(),
                              generator=torch.Generator().manual_seed(8888), return_dict=True, ddim_init_latents_t_idx=0,
                              output_type="latent").frames
        ref2._guidance_scale = 9.0
        out["rec_pe"], out["rec_npe"] = ref2.encode_prompt("", dev, 1, neg, clip_skip=1)
    ref.scheduler = fwd
    ref_pnp.register_conv_injection(ref, ts[: int(n_steps * ratios[0])])
    ref_pnp.register_spatial_attention_pnp(ref, ts[: int(n_steps * ratios[1])])
    ref_pnp.register_temp_attention_pnp(ref, ts[: int(n_steps * ratios[2])])
    out["edit_ref"] = ref.sample_with_pnp(
        prompt=edit_prompt, image=edited, height=size, width=size, num_frames=Fr, num_inference_steps=n_steps, guidance_scale=9.0,
        negative_prompt=neg, target_fps=8, latents=files[T].clone(), generator=torch.Generator().manual_seed(8888), return_dict=True,
        ddim_init_latents_t_idx=0, ddim_inv_latents_path=out_dir, ddim_inv_prompt="", ddim_inv_1st_frame=frames[0],
        output_type="latent").frames
    ref._guidance_scale = 9.0
    out["pe"], out["npe"] = ref.encode_prompt(edit_prompt, dev, 1, neg, clip_skip=1)
    out["ie2"] = ref._encode_image(crop(edited), dev, 1)                                                # [zeros, positive]
    out["il2"] = ref.prepare_image_latents(prep(edited), device=dev, num_frames=Fr, num_videos_per_prompt=1)  # [edited, edited]
    return out
