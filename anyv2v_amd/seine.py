"""The SEINE backend's model on the HIP kernels: the whole ``UNet3DConditionModel`` (``seine/models/unet.py:98-560``:
``CrossAttnDownBlock3D`` / ``DownBlock3D`` / ``UNetMidBlock3DCrossAttn`` / ``UpBlock3D`` / ``CrossAttnUpBlock3D`` of
``seine/models/unet_blocks.py``, each ``ResnetBlock3D`` + ``Transformer3DModel``) plus the five registration functions of
``seine/pnp_utils.py:121-458`` with the reference's names and arguments (``register_time``, ``register_conv_injection``,
``register_spatial_attention_pnp``, ``register_cross_attention_pnp`` -- a hook I2VGen-XL has no analogue of -- and
``register_temp_attention_pnp``).  Module tree and state-dict keys are the reference's (``seine.pt``'s ``ema`` dict loads strictly); the
runner classes and CLIs around it are ``anyv2v_amd/seine_pipeline.py`` and ``seine_run_*.py``.

What this family adds over I2VGen-XL, on the token layout ``X[(b f)(h w), C]``:

* one transformer block holds spatial self-attention, text cross-attention AND temporal self-attention (``attn1`` / ``attn2`` /
  ``attn_temp``, ``seine/models/attention.py:439-647``), the temporal one on the frame-strided view of the same tokens;
* ``attn_temp`` rotates the first 32 channels of EVERY head (``RotaryEmbedding(32)`` on [b, heads, f, d], ``attention.py:880-882``,
  ``unet.py:185``) and adds a learned relative-position bias [heads, F, F] to the scaled scores (``:815-817,887``):
  ``anyv2v_rotary_f16`` with one window per head, ``anyv2v_attention_bias_f16`` with the bias table built once per frame count;
* the cross-attention hook (``pnp_utils.py:302-376``) makes Q AND the text keys of the negative / editing branches the source
  branch's: ``qk_mod`` aliasing with the text K / V addressed through ``kv_div`` -- no copies;
* ``ResnetBlock3D`` normalises [b, c, f, h, w]: GroupNorm statistics over all frames of a batch element
  (``ResnetBlock2D.norm_over_frames``); its convolutions are per frame (``InflatedConv3d``).

The reference installs its hooks by replacing ``module.forward`` and setting ``t`` / ``injection_schedule`` on the ATTENTION MODULES
(not on processors); the native modules read the same two attributes."""
from __future__ import annotations

import types
from typing import List

import torch
from torch import nn

from . import ops
from .consisti2v import ROTARY_THETA, _Ctx, _RotaryFreqs, _sched
from .unet import (Conv2d, Downsample2D, FeedForward, GroupNorm, Identity, LayerNorm, Linear, ResnetBlock2D, SiLU, Upsample2D, pnp_on,
                   upsample_tokens)


class CrossAttention(nn.Module):
    """``seine/models/attention.py:43-312`` as the decoder uses it (no added KV projections, no relative positions): bias-free
    q / k / v, ``to_out = [Linear, Dropout]``.  ``t`` / ``injection_schedule``: the attributes the reference's hooks set."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64):
        super().__init__()
        self.inner_dim, self.heads, self.dim_head = heads * dim_head, heads, dim_head
        self.scale = dim_head ** -0.5
        self.is_cross = cross_attention_dim is not None
        kdim = cross_attention_dim if self.is_cross else query_dim
        self.to_q = Linear(query_dim, self.inner_dim, bias=False)
        self.to_k = Linear(kdim, self.inner_dim, bias=False)
        self.to_v = Linear(kdim, self.inner_dim, bias=False)
        self.to_out = nn.ModuleList([Linear(self.inner_dim, query_dim, bias=True), Identity()])
        self.group_norm = self.added_kv_proj_dim = None
        self.use_relative_position = False
        self.t = None
        self.injection_schedule = None
        self._w_qkv = self._w_kv = None

    def pack(self):
        if self.is_cross:
            self._w_kv = torch.cat([self.to_k.weight.data, self.to_v.weight.data], 0).contiguous()
        else:
            self._w_qkv = torch.cat([self.to_q.weight.data, self.to_k.weight.data, self.to_v.weight.data], 0).contiguous()

    def _out(self, o, residual):
        return ops.gemm(o, self.to_out[0].weight, bias=self.to_out[0].bias, residual=residual)

    def run_spatial(self, ctx, h, residual):
        """``attn1``: self-attention over the HW tokens of every frame; hooked form ``seine/pnp_utils.py:199-299``."""
        B, F, HW, C = ctx.B, ctx.F, ctx.H * ctx.W, self.inner_dim
        o = torch.empty((h.shape[0], C), dtype=torch.float16, device=h.device)
        qkv = ops.gemm(h, self._w_qkv)
        qk_mod = (B * F) // 3 if pnp_on(self.t, self.injection_schedule) else 0
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=B * F, heads=self.heads, Sq=HW, Sk=HW, q_strides=(HW, 0, 1),
                      kv_strides=(HW, 0, 1), qk_mod=qk_mod, scale=self.scale, head_dim=self.dim_head)
        return self._out(o, residual)

    def run_text(self, ctx, h, residual):
        """``attn2``: cross-attention to the L text tokens of the batch element (repeated per frame by ``Transformer3DModel``,
        ``attention.py:392-394``); hooked form ``seine/pnp_utils.py:302-376``: Q and the text KEYS of all branches from the source."""
        B, F, HW, C = ctx.B, ctx.F, ctx.H * ctx.W, self.inner_dim
        o = torch.empty((h.shape[0], C), dtype=torch.float16, device=h.device)
        q = ops.gemm(h, self.to_q.weight)
        kv = ops.gemm(ctx.context, self._w_kv)   # [B L, 2 C]
        qk_mod = (B * F) // 3 if pnp_on(self.t, self.injection_schedule) else 0
        ops.attention(q, kv[:, :C], kv[:, C:], o, batch=B * F, heads=self.heads, Sq=HW, Sk=ctx.L, q_strides=(HW, 0, 1),
                      kv_strides=(ctx.L, 0, 1), kv_div=F, qk_mod=qk_mod, scale=self.scale, head_dim=self.dim_head)
        return self._out(o, residual)


class _RelativePositionBias(nn.Module):
    """``seine/models/attention.py:930-967``: T5-style bucketed relative-position bias, an ``nn.Embedding(num_buckets, heads)``."""

    def __init__(self, heads, num_buckets=32, max_distance=32):
        super().__init__()
        self.num_buckets, self.max_distance = num_buckets, max_distance
        self.relative_attention_bias = nn.Embedding(num_buckets, heads)

    def table(self, n, device):
        """[heads, n, n] fp32, entry (h, i, j) = embedding[bucket(j - i), h] (``:943-967``)."""
        import math
        pos = torch.arange(n, dtype=torch.long)
        rel = pos[None, :] - pos[:, None]             # k_pos - q_pos
        nb = self.num_buckets // 2
        m = -rel
        ret = (m < 0).long() * nb
        m = m.abs()
        max_exact = nb // 2
        large = max_exact + (torch.log(m.float().clamp(min=1) / max_exact) / math.log(self.max_distance / max_exact) * (nb - max_exact)).long()
        large = torch.minimum(large, torch.full_like(large, nb - 1))
        bucket = ret + torch.where(m < max_exact, m, large)
        w = self.relative_attention_bias.weight.detach().float().cpu()
        return w[bucket].permute(2, 0, 1).contiguous().to(device)


class TemporalAttention(CrossAttention):
    """``seine/models/attention.py:797-927``: self-attention over the F frames of every (batch element, pixel) with per-head rotary
    position embedding and the relative-position bias; hooked form ``seine/pnp_utils.py:378-458``."""

    ROT = 32   # unet.py:185: RotaryEmbedding(32), shared by every block

    def __init__(self, query_dim, heads=8, dim_head=64):
        super().__init__(query_dim, None, heads, dim_head)
        self.time_rel_pos_bias = _RelativePositionBias(heads)
        self.rotary_emb = _RotaryFreqs(self.ROT)
        self._bias = {}

    def pack(self):
        super().pack()
        self._bias = {}
        assert self.dim_head >= self.ROT and self.dim_head % 8 == 0, "rotary_embedding: head_dim must be >= 32 (reference assertion)"
        want = 1.0 / (ROTARY_THETA ** (torch.arange(0, self.ROT, 2)[: self.ROT // 2].float() / self.ROT))
        if not torch.allclose(self.rotary_emb.freqs.detach().float().cpu(), want, rtol=1e-4, atol=0):
            raise NotImplementedError("rotary frequencies other than theta = 10000 ('lang') are not supported by anyv2v_rotary_f16")

    def run_temporal(self, ctx, h, residual):
        B, F, HW, C = ctx.B, ctx.F, ctx.H * ctx.W, self.inner_dim
        o = torch.empty((h.shape[0], C), dtype=torch.float16, device=h.device)
        qkv = ops.gemm(h, self._w_qkv)
        for col0 in (0, C):   # Q and K: the first 32 channels of every head, position = frame index
            ops.rotary(qkv, col0, self.ROT, HW, F, ROTARY_THETA, windows=self.heads, window_stride=self.dim_head)
        key = (F, str(h.device))
        if key not in self._bias:
            self._bias[key] = self.time_rel_pos_bias.table(F, h.device)
        qs = (F * HW, 1, HW)
        qk_mod = (B * HW) // 3 if pnp_on(self.t, self.injection_schedule) else 0
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=B * HW, heads=self.heads, Sq=F, Sk=F, inner=HW, q_strides=qs,
                      kv_strides=qs, qk_mod=qk_mod, scale=self.scale, head_dim=self.dim_head, bias=self._bias[key])
        return self._out(o, residual)


class BasicTransformerBlock(nn.Module):
    """``seine/models/attention.py:439-647`` (inference path): norm1 -> attn1 -> norm2 -> attn2 -> norm_temp -> attn_temp -> norm3 ->
    GEGLU feed-forward, each with a residual."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, heads, dim_head)
        self.norm1 = LayerNorm(dim)
        self.attn2 = CrossAttention(dim, cross_attention_dim, heads, dim_head)
        self.norm2 = LayerNorm(dim)
        self.attn_temp = TemporalAttention(dim, heads, dim_head)
        self.norm_temp = LayerNorm(dim)
        self.ff = FeedForward(dim)
        self.norm3 = LayerNorm(dim)

    def run(self, ctx, x):
        ln = lambda n, v: ops.layernorm(v, n.weight, n.bias, n.eps)
        x = self.attn1.run_spatial(ctx, ln(self.norm1, x), x)
        x = self.attn2.run_text(ctx, ln(self.norm2, x), x)
        x = self.attn_temp.run_temporal(ctx, ln(self.norm_temp, x), x)
        return self.ff.run(ln(self.norm3, x), x)


class Transformer3DModel(nn.Module):
    """``seine/models/attention.py:314-436``: per-frame GroupNorm(eps 1e-6) -> proj_in -> blocks -> proj_out -> + input."""

    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups, use_linear_projection=False):
        super().__init__()
        inner = heads * dim_head
        self.norm = GroupNorm(groups, in_channels, eps=1e-6)
        mk = (lambda a, b: Linear(a, b)) if use_linear_projection else (lambda a, b: Conv2d(a, b, 1))
        self.proj_in = mk(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = mk(inner, in_channels)

    def run(self, ctx, x):
        h = ops.groupnorm(x, self.norm.weight, self.norm.bias, ctx.stats, ctx.H * ctx.W, groups=self.norm.num_groups, eps=self.norm.eps)
        h = ops.gemm(h, self.proj_in.weight.reshape(self.proj_in.weight.shape[0], -1), bias=self.proj_in.bias)
        for blk in self.transformer_blocks:
            h = blk.run(ctx, h)
        return ops.gemm(h, self.proj_out.weight.reshape(self.proj_out.weight.shape[0], -1), bias=self.proj_out.bias, residual=x)


def _res3d(cin, cout, temb_channels, groups, eps):
    r = ResnetBlock2D(cin, cout, temb_channels, groups, eps)
    r.norm_over_frames = True     # ResnetBlock3D: GroupNorm statistics over all frames of a batch element
    return r


class _SeineBlock(nn.Module):
    """Packing (one GEMM for the time-embedding projections of the block's ResNets) shared by the block types below."""

    def pack(self):
        for m in self.modules():
            if m is not self and hasattr(m, "pack"):
                m.pack()
        col = 0
        for r in self.resnets:
            r._temb_col = col
            col += r.out_channels
        self._w_temb = torch.cat([r.time_emb_proj.weight.data for r in self.resnets], 0).contiguous()
        self._b_temb = torch.cat([r.time_emb_proj.bias.data for r in self.resnets], 0).contiguous()
        self._packed = True

    def enter(self, ctx):
        ctx.temb_all = ops.gemm(ops.silu(ctx.emb), self._w_temb, bias=self._b_temb)


class CrossAttnUpBlock3D(_SeineBlock):
    """``seine/models/unet_blocks.py:444-575``; constructor argument names are the reference's."""

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 attn_num_head_channels=1, cross_attention_dim=1280, add_upsample=True, use_linear_projection=False,
                 use_first_frame=False, use_relative_position=False, rotary_emb=None, **unused):
        super().__init__()
        if use_first_frame or use_relative_position:
            raise NotImplementedError("native CrossAttnUpBlock3D: use_first_frame / use_relative_position are off in the released model")
        self.has_cross_attention = True
        self.attn_num_head_channels, self.groups = attn_num_head_channels, resnet_groups
        self.resnets, self.attentions = nn.ModuleList(), nn.ModuleList()
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            r = ResnetBlock2D(rin + skip, out_channels, temb_channels, resnet_groups, resnet_eps)
            r.norm_over_frames = True
            self.resnets.append(r)
            self.attentions.append(Transformer3DModel(attn_num_head_channels, out_channels // attn_num_head_channels, out_channels,
                                                      cross_attention_dim, resnet_groups, use_linear_projection))
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None
        self._packed = False
        self._w_temb = self._b_temb = None

    def load_state_dict(self, sd, strict=True, **kw):
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._packed = False
        return out

    def run(self, ctx, x, skips: List[torch.Tensor], out_hw=None):
        """``out_hw``: ``upsample_size`` of the reference (``unet_blocks.py:530,572``): the size of the skip connections the next block pops."""
        for resnet, attn in zip(self.resnets, self.attentions):
            x = resnet.run(ctx, x, skips.pop(), ctx.H, ctx.W)
            x = attn.run(ctx, x)
        if self.upsamplers is not None:
            x, Ho, Wo = upsample_tokens(self.upsamplers[0].conv, x, ctx.H, ctx.W, out_hw)
            ctx.set_hw(Ho, Wo)
        return x

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None, upsample_size=None, **unused):
        """The reference's call (``unet_blocks.py:524-575``): ``hidden_states`` and the skip tensors [b, c, f, h, w], ``temb`` [b, D],
        ``encoder_hidden_states`` [b, L, D]; ``upsample_size`` = (f, h, w) or (h, w) of the next block's skip connections."""
        if not self._packed:
            self.pack()
        B, C, F, H, W = hidden_states.shape
        ctx = _Ctx(B, F, H, W, hidden_states.device, self.groups)

        def tok(t):
            return t.to(torch.float16).permute(0, 2, 3, 4, 1).reshape(B * F * H * W, t.shape[1]).contiguous()
        emb = temb.to(torch.float16).contiguous()
        ctx.emb = emb
        ctx.temb_all = ops.gemm(ops.silu(emb), self._w_temb, bias=self._b_temb)
        ehs = encoder_hidden_states.to(torch.float16)
        ctx.L = ehs.shape[1]
        ctx.context = ehs.reshape(B * ctx.L, -1).contiguous()
        y = self.run(ctx, tok(hidden_states), [tok(s) for s in res_hidden_states_tuple],
                     out_hw=None if upsample_size is None else tuple(int(v) for v in upsample_size[-2:]))
        return y.view(B, F, ctx.H, ctx.W, -1).permute(0, 4, 1, 2, 3)


# ------------------------------------------------------------------------------------------------- the other blocks, the UNet
def _seine_opts(use_first_frame, use_relative_position):
    if use_first_frame or use_relative_position:
        raise NotImplementedError("use_first_frame / use_relative_position are off in the released SEINE model")


class CrossAttnDownBlock3D(_SeineBlock):
    """``seine/models/unet_blocks.py:235-362``: per layer ResnetBlock3D -> Transformer3DModel; stride-2 ``Downsample3D`` behind."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, attn_num_head_channels=1,
                 cross_attention_dim=1280, add_downsample=True, use_linear_projection=False, use_first_frame=False,
                 use_relative_position=False, **unused):
        super().__init__()
        _seine_opts(use_first_frame, use_relative_position)
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([_res3d(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_groups, resnet_eps)
                                      for i in range(num_layers)])
        self.attentions = nn.ModuleList([Transformer3DModel(attn_num_head_channels, out_channels // attn_num_head_channels, out_channels,
                                                            cross_attention_dim, resnet_groups, use_linear_projection)
                                         for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def run(self, ctx, x):
        outs = []
        for resnet, attn in zip(self.resnets, self.attentions):
            x = attn.run(ctx, resnet.run(ctx, x, None, ctx.H, ctx.W))
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].conv.tokens(x, ctx.H, ctx.W)
            ctx.set_hw((ctx.H - 1) // 2 + 1, (ctx.W - 1) // 2 + 1)     # (stride 2, padding 1: odd sizes round up)
            outs.append(x)
        return x, outs


class DownBlock3D(_SeineBlock):
    """``seine/models/unet_blocks.py:365-441``."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, add_downsample=True, **unused):
        super().__init__()
        self.has_cross_attention = False
        self.resnets = nn.ModuleList([_res3d(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_groups, resnet_eps)
                                      for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def run(self, ctx, x):
        outs = []
        for resnet in self.resnets:
            x = resnet.run(ctx, x, None, ctx.H, ctx.W)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].conv.tokens(x, ctx.H, ctx.W)
            ctx.set_hw((ctx.H - 1) // 2 + 1, (ctx.W - 1) // 2 + 1)     # (stride 2, padding 1: odd sizes round up)
            outs.append(x)
        return x, outs


class UNetMidBlock3DCrossAttn(_SeineBlock):
    """``seine/models/unet_blocks.py:145-232``: resnet, then (transformer, resnet) per layer."""

    def __init__(self, in_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, attn_num_head_channels=1,
                 cross_attention_dim=1280, use_linear_projection=False, use_first_frame=False, use_relative_position=False, **unused):
        super().__init__()
        _seine_opts(use_first_frame, use_relative_position)
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([_res3d(in_channels, in_channels, temb_channels, resnet_groups, resnet_eps) for _ in range(num_layers + 1)])
        self.attentions = nn.ModuleList([Transformer3DModel(attn_num_head_channels, in_channels // attn_num_head_channels, in_channels,
                                                            cross_attention_dim, resnet_groups, use_linear_projection)
                                         for _ in range(num_layers)])

    def run(self, ctx, x):
        x = self.resnets[0].run(ctx, x, None, ctx.H, ctx.W)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            x = resnet.run(ctx, attn.run(ctx, x), None, ctx.H, ctx.W)
        return x


class UpBlock3D(_SeineBlock):
    """``seine/models/unet_blocks.py:577-648``."""

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 add_upsample=True, **unused):
        super().__init__()
        self.has_cross_attention = False
        self.resnets = nn.ModuleList()
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            self.resnets.append(_res3d(rin + skip, out_channels, temb_channels, resnet_groups, resnet_eps))
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def run(self, ctx, x, skips: List[torch.Tensor], out_hw=None):
        for resnet in self.resnets:
            x = resnet.run(ctx, x, skips.pop(), ctx.H, ctx.W)
        if self.upsamplers is not None:
            x, Ho, Wo = upsample_tokens(self.upsamplers[0].conv, x, ctx.H, ctx.W, out_hw)
            ctx.set_hw(Ho, Wo)
        return x


class _TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = Linear(in_channels, time_embed_dim)
        self.linear_2 = Linear(time_embed_dim, time_embed_dim)

    def run(self, x):
        return ops.gemm(ops.gemm(x, self.linear_1.weight, bias=self.linear_1.bias, act=ops.ACT_SILU), self.linear_2.weight,
                        bias=self.linear_2.bias)


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class UNet3DConditionModel(nn.Module):
    """``seine/models/unet.py:98-560`` on the HIP kernels, for the released configuration family (Stable-Diffusion-1.x layout inflated to
    video; ``use_concat``: ``in_channels`` 9 = noisy latents | mask | masked-video latents; no class embedding, ``use_first_frame`` /
    ``use_relative_position`` off).  Module tree and state-dict keys are the reference's (the ``ema`` state dict of ``seine.pt`` loads
    strictly).  ``forward(sample [B, 9, F, h, w], timestep, encoder_hidden_states [B, L, D]).sample`` -> [B, 4, F, h, w]."""

    def __init__(self, sample_size=None, in_channels=4, out_channels=4, flip_sin_to_cos=True, freq_shift=0,
                 down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                 mid_block_type="UNetMidBlock3DCrossAttn",
                 up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=1280,
                 attention_head_dim=8, use_linear_projection=False, class_embed_type=None, num_class_embeds=None, use_first_frame=False,
                 use_relative_position=False, **unused):
        super().__init__()
        if not flip_sin_to_cos or freq_shift != 0 or class_embed_type is not None or num_class_embeds is not None \
                or mid_block_type != "UNetMidBlock3DCrossAttn":
            raise NotImplementedError("native UNet3DConditionModel: flip_sin_to_cos=True, freq_shift=0, no class embedding only")
        _seine_opts(use_first_frame, use_relative_position)
        nb = len(block_out_channels)
        heads = tuple(attention_head_dim) if isinstance(attention_head_dim, (list, tuple)) else (attention_head_dim,) * nb
        boc = tuple(block_out_channels)
        self.config = _Cfg(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels, block_out_channels=boc,
                           cross_attention_dim=cross_attention_dim, center_input_sample=False, class_embed_type=None)
        self.sample_size, self.groups = sample_size, norm_num_groups
        ted = boc[0] * 4
        self.conv_in = Conv2d(in_channels, boc[0], 3, padding=1, pad_cin_to=64)
        self.time_embedding = _TimestepEmbedding(boc[0], ted)
        common = dict(temb_channels=ted, resnet_eps=norm_eps, resnet_groups=norm_num_groups, cross_attention_dim=cross_attention_dim,
                      use_linear_projection=use_linear_projection)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, typ in enumerate(down_block_types):
            cin, out = out, boc[i]
            kw = dict(in_channels=cin, out_channels=out, num_layers=layers_per_block, add_downsample=i != nb - 1, **common)
            if typ == "CrossAttnDownBlock3D":
                self.down_blocks.append(CrossAttnDownBlock3D(attn_num_head_channels=heads[i], **kw))
            elif typ == "DownBlock3D":
                self.down_blocks.append(DownBlock3D(**kw))
            else:
                raise NotImplementedError(typ)
        self.mid_block = UNetMidBlock3DCrossAttn(in_channels=boc[-1], attn_num_head_channels=heads[-1], **common)
        self.up_blocks = nn.ModuleList()
        rboc, rheads = boc[::-1], heads[::-1]
        out = rboc[0]
        for i, typ in enumerate(up_block_types):
            prev, out = out, rboc[i]
            kw = dict(in_channels=rboc[min(i + 1, nb - 1)], out_channels=out, prev_output_channel=prev, num_layers=layers_per_block + 1,
                      add_upsample=i != nb - 1, **common)
            if typ == "CrossAttnUpBlock3D":
                self.up_blocks.append(CrossAttnUpBlock3D(attn_num_head_channels=rheads[i], **kw))
            elif typ == "UpBlock3D":
                self.up_blocks.append(UpBlock3D(**kw))
            else:
                raise NotImplementedError(typ)
        self.conv_norm_out = GroupNorm(norm_num_groups, boc[0], norm_eps)
        self.conv_act = SiLU()
        self.conv_out = Conv2d(boc[0], out_channels, 3, padding=1)
        self._packed = False

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def pack(self):
        for blk in list(self.down_blocks) + [self.mid_block] + list(self.up_blocks):
            blk.pack()
        self.conv_in.pack()
        self.conv_out.pack()
        self._packed = True

    def load_state_dict(self, sd, strict=True, **kw):
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._packed = False
        return out

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states=None, return_dict=True, **unused):
        if not self._packed:
            self.pack()
        dev = sample.device
        B, C, F, H, W = sample.shape
        ctx = _Ctx(B, F, H, W, dev, self.groups)
        c0 = self.config.block_out_channels[0]
        t = torch.as_tensor(timestep, device=dev).reshape(-1).float().expand(B).contiguous()
        ctx.emb = self.time_embedding.run(ops.timestep_embedding(t, c0))
        ehs = encoder_hidden_states.to(torch.float16)
        ctx.L = ehs.shape[1]
        ctx.context = ehs.reshape(B * ctx.L, -1).contiguous()
        xin = torch.zeros((B * F * H * W, 64), dtype=torch.float16, device=dev)
        ops.ncfhw_to_tokens(sample.to(torch.float16).contiguous(), xin, col0=0)
        x = self.conv_in.tokens(xin, H, W)
        skips = [x]
        sizes = [(H, W)]     # per resolution level: the up path returns to exactly these (``upsample_size``, ``unet.py:393-401,485-500``)
        for blk in self.down_blocks:
            blk.enter(ctx)
            x, outs = blk.run(ctx, x)
            skips.extend(outs)
            if blk.downsamplers is not None:
                sizes.append((ctx.H, ctx.W))
        self.mid_block.enter(ctx)
        x = self.mid_block.run(ctx, x)
        for blk in self.up_blocks:
            blk.enter(ctx)
            sizes.pop()
            x = blk.run(ctx, x, skips, out_hw=sizes[-1] if sizes else None)
        # (conv_norm_out is a plain GroupNorm on [b, c, f, h, w]: statistics over all frames of a batch element)
        x = ops.groupnorm(x, self.conv_norm_out.weight, self.conv_norm_out.bias, ctx.stats, F * H * W, groups=self.groups,
                          eps=self.conv_norm_out.eps, silu=True)
        vtok = torch.empty((x.shape[0], 8), dtype=torch.float16, device=dev)
        self.conv_out.tokens(x, H, W, out=vtok)
        out = ops.tokens_to_ncfhw(vtok, B, self.config.out_channels, F, H, W)
        if not return_dict:
            return (out,)
        return types.SimpleNamespace(sample=out)


# ------------------------------------------------------------------------------------------------- hook registration
_UP = {1: [0, 1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}      # seine/pnp_utils.py:125
_DOWN = {0: [0, 1], 1: [0, 1], 2: [0, 1]}             # :124
_INJ = {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}        # :287, :368, :453


def _blocks_of(unet):
    for m in unet.modules():
        if isinstance(m, BasicTransformerBlock):
            yield m


def register_time(model, t):
    """``seine/pnp_utils.py:121-147``: the conv site and attn1 / attn2 / attn_temp of every up / down / mid transformer block."""
    t = int(t)
    setattr(model.unet.up_blocks[1].resnets[1], "t", t)
    sites = [model.unet.up_blocks[res].attentions[b] for res, bs in _UP.items() for b in bs]
    down = getattr(model.unet, "down_blocks", None)
    if down is not None:
        sites += [down[res].attentions[b] for res, bs in _DOWN.items() for b in bs]
    if getattr(model.unet, "mid_block", None) is not None:
        sites.append(model.unet.mid_block.attentions[0])
    for site in sites:
        blk = site.transformer_blocks[0]
        for name in ("attn1", "attn2", "attn_temp"):
            setattr(getattr(blk, name), "t", t)


def clear_time(model):
    """No hook fires until the next ``register_time`` (the reference's inversion runs in a process of its own, without hooks)."""
    model.unet.up_blocks[1].resnets[1].t = None
    for blk in _blocks_of(model.unet):
        for name in ("attn1", "attn2", "attn_temp"):
            getattr(blk, name).t = None


def register_conv_injection(model, injection_schedule, d_s=0.1, d_t=0.5):
    """``seine/pnp_utils.py:150-196``."""
    setattr(model.unet.up_blocks[1].resnets[1], "injection_schedule", _sched(injection_schedule))


def _register(model, name, injection_schedule):
    for blk in _blocks_of(model.unet):   # "Disable PNP" on every block first (:282-285), then the schedule on decoder blocks 4-11
        setattr(getattr(blk, name), "injection_schedule", frozenset())
    for res, bs in _INJ.items():
        for b in bs:
            setattr(getattr(model.unet.up_blocks[res].attentions[b].transformer_blocks[0], name), "injection_schedule",
                    _sched(injection_schedule))


def register_spatial_attention_pnp(model, injection_schedule, d_s=0.1, d_t=0.5):
    """``seine/pnp_utils.py:199-299``."""
    _register(model, "attn1", injection_schedule)


def register_cross_attention_pnp(model, injection_schedule):
    """``seine/pnp_utils.py:302-376``."""
    _register(model, "attn2", injection_schedule)


def register_temp_attention_pnp(model, injection_schedule, d_s=0.1, d_t=0.5):
    """``seine/pnp_utils.py:378-458``."""
    _register(model, "attn_temp", injection_schedule)
