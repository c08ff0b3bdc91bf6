"""The ConsistI2V backend's model on the HIP kernels: the whole ``VideoLDMUNet3DConditionModel``
(``consisti2v/consisti2v/models/videoldm_unet.py:68-1064``) -- encoder / mid / decoder blocks of
``videoldm_unet_blocks.py:225-1158`` with their ``ResnetBlock2D`` / ``TemporalResnetBlock`` / spatial and temporal
``Transformer2DConditionModel`` layers, time and frame-stride embeddings, first-frame conditioning by concatenation -- plus the four
hook registration functions of ``consisti2v/pnp_utils.py:19-345`` with the reference's names and arguments (the hook sites are all in
``unet.up_blocks[1..3]``).  Module tree and state-dict keys are the reference's, so a real checkpoint loads unchanged; the pipeline and
the runners around it are ``anyv2v_amd/consisti2v_pipeline.py`` and ``consisti2v_run_*.py``.

What differs from the I2VGen-XL family (``anyv2v_amd/unet.py``) and how it is computed here, all on the token layout
``X[(b f)(h w), C]``:

* spatial ``attn1`` attends over [own frame ; first frame of the clip] (``videoldm_transformer_blocks.py:479-489``): K and V are
  per-token projections, so the first-frame half IS the K / V rows of frame 0 -- one fused QKV GEMM, one row gather that lays the
  two halves side by side, the flash kernel with Sk = 2 HW.  PnP injection (``consisti2v/pnp_utils.py:188-197``) aliases Q and K
  of all three branches to the source branch inside the kernel (qk_mod), as for I2VGen-XL.
* temporal ``attn1`` attends over [the pixel's F frames ; the pixel's 8 neighbours in the first frame] with rotary position
  embedding on the first half of the channels (``:490-503``, ``videoldm_attention.py:589-599,773-777``).  The rotary kernel runs in
  place on the Q and K columns of the fused projection with the frame index as position; the neighbour keys have position 0
  (``key_pos_idx``), where the rotation is the identity -- they are rows of frame 0 of the same K / V, fetched by the row gather.
  Injection happens before the rotation in the reference (``pnp_utils.py:296-310``); the rotation depends on the position only,
  so aliasing after it is the same thing.
* temporal ``attn2`` (text cross-attention, ``RotaryEmbAttnProcessor2_0``): only Q is rotated (qlen != klen).
* ``TemporalResnetBlock`` and the temporal transformer blend with a learned ``alpha``: alpha x + (1 - alpha)(x + f(x)) =
  x + (1 - alpha) f(x); (1 - alpha) is folded into the last projection's weights at pack time.
* GroupNorm of the temporal layers is the 4-D one (per frame), eps 1e-6.

Temporal attention has head_dim C / 8 = 40 / 80 / 160: the whole-sequence MFMA kernel (``small_attn_mfma_kernel``, 16 + 8 keys).
Not tuned beyond that: no HIP-graph gain for this family (measured), no source-branch shortcuts.
"""
from __future__ import annotations

import types
from typing import Dict, List

import torch
from torch import nn

from . import ops
from .ops import MODE_TEMPORAL
from .unet import (Conv2d, Conv3dTemporal, Downsample2D, FeedForward, GroupNorm, Identity, LayerNorm, Linear, ResnetBlock2D, SiLU,
                   Upsample2D, pnp_on, upsample_tokens)

ROTARY_THETA = 10000.0  # rotary_embedding.py:76 (freqs_for="lang")


class _Ctx:
    """Per-call state of a block: geometry, GroupNorm scratch, this block's time-embedding projections, the text tokens."""

    def __init__(self, B, F, H, W, device, groups):
        self.B, self.F, self.H, self.W = B, F, H, W
        # (TemporalResnetBlock keeps the reference's default of 32 groups whatever ``resnet_groups`` is, videoldm_unet_blocks.py:231)
        self.stats = torch.empty(ops.gn_scratch_floats(B * F, 1, max(groups, 32)), dtype=torch.float32, device=device)
        self.temb_all = None
        self.emb = None
        self.context = None   # [B * L, D] text tokens, one copy per batch element
        self.L = 0
        self._idx: Dict[tuple, torch.Tensor] = {}

    def set_hw(self, H, W):
        """The block stack moves to another resolution level (index tensors are cached per (B, F, H, W))."""
        self.H, self.W = H, W


def _first_frame_index(B, F, HW, device):
    """Row index of the key / value sequence [own frame ; first frame] of every image: out row ((n, half, s)) -> qkv row."""
    n = torch.arange(B * F, device=device)
    own = n[:, None] * HW + torch.arange(HW, device=device)[None]
    first = ((n // F) * F)[:, None] * HW + torch.arange(HW, device=device)[None]
    return torch.stack([own, first], 1).reshape(-1).to(torch.int32).contiguous()


def _window_index(B, F, H, W, device):
    """Row index of the key / value sequence of every (b, pixel): its F frames, then its 8 neighbours in frame 0 in the order of
    ``first_frame_windows[..., mask]`` (``videoldm_transformer_blocks.py:494-497``: 3 x 3 window of the replicate-padded frame,
    row-major, centre removed)."""
    HW = H * W
    y, x = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
    nb = []
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dy == 0 and dx == 0:
                continue
            nb.append(((y + dy).clamp(0, H - 1) * W + (x + dx).clamp(0, W - 1)).reshape(-1))
    nb = torch.stack(nb, 1)                                                    # [HW, 8] pixel index inside frame 0
    p = torch.arange(HW, device=device)
    b = torch.arange(B, device=device)
    frames = (b[:, None, None] * F + torch.arange(F, device=device)[None, None, :]) * HW + p[None, :, None]   # [B, HW, F]
    window = (b[:, None, None] * F * HW) + nb[None]                            # [B, HW, 8]
    return torch.cat([frames, window], 2).reshape(-1).to(torch.int32).contiguous()


# ------------------------------------------------------------------------------------------------- attention
class HipSpaAttnProcessor:
    """Native ``ModifiedSpaAttnProcessor`` (``consisti2v/pnp_utils.py:133-225``) / ``AttnProcessor2_0`` of a spatial
    ``ConditionalAttention``; ``injection_schedule`` / ``t`` as in the reference."""

    def __init__(self, injection_schedule=None):
        self.injection_schedule = injection_schedule
        self.t = None

    def run(self, attn, ctx, h, residual, first_frame: bool, kv=None):
        B, F, HW, C = ctx.B, ctx.F, ctx.H * ctx.W, attn.inner_dim
        T = h.shape[0]
        o = torch.empty((T, C), dtype=torch.float16, device=h.device)
        if kv is not None:   # text cross-attention: every frame of batch element b reads the same L text tokens
            q = ops.gemm(h, attn.to_q.weight)
            k_, v_ = kv
            ops.attention(q, k_, v_, o, batch=B * F, heads=attn.heads, Sq=HW, Sk=ctx.L, q_strides=(HW, 0, 1), kv_strides=(ctx.L, 0, 1),
                          kv_div=F, scale=attn.scale, head_dim=attn.dim_head)
            return ops.gemm(o, attn.to_out[0].weight, bias=attn.to_out[0].bias, residual=residual)
        inject = pnp_on(self.t, self.injection_schedule)
        qkv = ops.gemm(h, attn._w_qkv)
        qk_mod = (B * F) // 3 if inject else 0
        if first_frame:
            key = ("ff", B, F, HW)
            if key not in ctx._idx:
                ctx._idx[key] = _first_frame_index(B, F, HW, h.device)
            kvc = torch.empty((2 * T, 2 * C), dtype=torch.float16, device=h.device)
            ops.gather_rows(qkv, C, ctx._idx[key], kvc, 0, 2 * C)
            ops.attention(qkv[:, :C], kvc[:, :C], kvc[:, C:], o, batch=B * F, heads=attn.heads, Sq=HW, Sk=2 * HW, q_strides=(HW, 0, 1),
                          kv_strides=(2 * HW, 0, 1), qk_mod=qk_mod, scale=attn.scale, head_dim=attn.dim_head)
        else:
            ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=B * F, heads=attn.heads, Sq=HW, Sk=HW,
                          q_strides=(HW, 0, 1), kv_strides=(HW, 0, 1), qk_mod=qk_mod, scale=attn.scale, head_dim=attn.dim_head)
        return ops.gemm(o, attn.to_out[0].weight, bias=attn.to_out[0].bias, residual=residual)


class HipTmpAttnProcessor:
    """Native ``ModifiedTmpAttnProcessor`` (``consisti2v/pnp_utils.py:229-345``) / ``RotaryEmbAttnProcessor2_0``
    (``videoldm_attention.py:710-808``) of a ``TemporalConditionalAttention`` with rotary position embedding."""

    def __init__(self, injection_schedule=None):
        self.injection_schedule = injection_schedule
        self.t = None

    def run(self, attn, ctx, h, residual, adjacent: bool, kv=None):
        B, F, H, W, C = ctx.B, ctx.F, ctx.H, ctx.W, attn.inner_dim
        HW = H * W
        T = h.shape[0]
        o = torch.empty((T, C), dtype=torch.float16, device=h.device)
        qs = (F * HW, 1, HW)   # sequence of (b, pixel): F rows, HW apart
        if kv is not None:     # text cross-attention; qlen != klen: only the queries are rotated (videoldm_attention.py:773-777)
            q = ops.gemm(h, attn.to_q.weight)
            ops.rotary(q, 0, attn.rot_dim, HW, F, ROTARY_THETA)
            k_, v_ = kv
            # (the K / V batch index i // kv_div = b is decomposed with the same ``inner`` as the queries': b = (b // HW) HW + b % HW)
            ops.attention(q, k_, v_, o, batch=B * HW, heads=attn.heads, Sq=F, Sk=ctx.L, inner=HW, q_strides=qs,
                          kv_strides=(HW * ctx.L, ctx.L, 1), kv_div=HW, scale=attn.scale, head_dim=attn.dim_head)
            return ops.gemm(o, attn.to_out[0].weight, bias=attn.to_out[0].bias, residual=residual)
        inject = pnp_on(self.t, self.injection_schedule)
        qkv = ops.gemm(h, attn._w_qkv)
        ops.rotary(qkv, 0, attn.rot_dim, HW, F, ROTARY_THETA)   # Q
        ops.rotary(qkv, C, attn.rot_dim, HW, F, ROTARY_THETA)   # K (frame 0: angle 0, unchanged -- the neighbour keys' position)
        qk_mod = (B * HW) // 3 if inject else 0
        if adjacent:
            key = ("win", B, F, H, W)
            if key not in ctx._idx:
                ctx._idx[key] = _window_index(B, F, H, W, h.device)
            S = F + 8
            kvc = torch.empty((B * HW * S, 2 * C), dtype=torch.float16, device=h.device)
            ops.gather_rows(qkv, C, ctx._idx[key], kvc, 0, 2 * C)
            ops.attention(qkv[:, :C], kvc[:, :C], kvc[:, C:], o, batch=B * HW, heads=attn.heads, Sq=F, Sk=S, inner=HW, q_strides=qs,
                          kv_strides=(HW * S, S, 1), qk_mod=qk_mod, scale=attn.scale, head_dim=attn.dim_head)
        else:
            ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=B * HW, heads=attn.heads, Sq=F, Sk=F, inner=HW,
                          q_strides=qs, kv_strides=qs, qk_mod=qk_mod, scale=attn.scale, head_dim=attn.dim_head)
        return ops.gemm(o, attn.to_out[0].weight, bias=attn.to_out[0].bias, residual=residual)


class ConditionalAttention(nn.Module):
    """``videoldm_attention.py:49-176`` (the fields the decoder uses): bias-free q / k / v, ``to_out = [Linear, Dropout]``."""

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
        self.spatial_norm = self.group_norm = self.norm_cross = None
        self.residual_connection, self.rescale_output_factor = False, 1.0
        self.processor = HipSpaAttnProcessor()
        self._w_qkv = self._w_kv = None

    def pack(self):
        if self.is_cross:
            self._w_kv = torch.cat([self.to_k.weight.data, self.to_v.weight.data], 0).contiguous()
        else:
            self._w_qkv = torch.cat([self.to_q.weight.data, self.to_k.weight.data, self.to_v.weight.data], 0).contiguous()

    def text_kv(self, ctx):
        kv = ops.gemm(ctx.context, self._w_kv)   # [B L, 2 C]
        return kv[:, :self.inner_dim], kv[:, self.inner_dim:]


class _RotaryFreqs(nn.Module):
    """``RotaryEmbedding.freqs`` (``rotary_embedding.py:88-100``): a (non-learned) Parameter in the reference's state dict."""

    def __init__(self, dim):
        super().__init__()
        self.freqs = nn.Parameter(1.0 / (ROTARY_THETA ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim)), requires_grad=False)


class _RelativePositionBias(nn.Module):
    """``videoldm_attention.py:668-707``: constructed by the reference, never used by its forward (commented out at :765-768)."""

    def __init__(self, heads, num_buckets=32):
        super().__init__()
        self.relative_attention_bias = nn.Embedding(num_buckets, heads)


class TemporalConditionalAttention(ConditionalAttention):
    """``videoldm_attention.py:552-641`` with ``rotary_emb=True`` (the released model's ``temp_pos_embedding: rotary``)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, n_frames=16):
        super().__init__(query_dim, cross_attention_dim, heads, dim_head)
        self.n_frames = n_frames
        self.use_rotary_emb = True
        self.rot_dim = self.inner_dim // 2           # RotaryEmbedding(self.inner_dim // 2), applied before the head split
        self.rotary_emb = _RotaryFreqs(self.rot_dim)
        self.rotary_bias = _RelativePositionBias(heads)
        self.processor = HipTmpAttnProcessor()

    def pack(self):
        super().pack()
        want = 1.0 / (ROTARY_THETA ** (torch.arange(0, self.rot_dim, 2)[: self.rot_dim // 2].float() / self.rot_dim))
        if not torch.allclose(self.rotary_emb.freqs.detach().float().cpu(), want, rtol=1e-4, atol=0):
            raise NotImplementedError("rotary frequencies other than theta = 10000 ('lang') are not supported by anyv2v_rotary_f16")
        assert self.rot_dim % 8 == 0, "rotary window must be a multiple of 8 channels"


# ------------------------------------------------------------------------------------------------- transformer blocks
class BasicConditionalTransformerBlock(nn.Module):
    """``videoldm_transformer_blocks.py:321-563``: norm1 -> attn1 (+first frame / +adjacent first-frame window) -> norm2 -> attn2
    (text) -> norm3 -> GEGLU feed-forward, each with a residual."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim, n_frames, is_temporal, augment_temporal_attention):
        super().__init__()
        self.n_frames, self.is_temporal, self.augment_temporal_attention = n_frames, is_temporal, augment_temporal_attention
        self.only_cross_attention = False
        A = (lambda **kw: TemporalConditionalAttention(n_frames=n_frames, **kw)) if is_temporal else ConditionalAttention
        self.norm1 = LayerNorm(dim)
        self.attn1 = A(query_dim=dim, heads=heads, dim_head=dim_head)
        self.norm2 = LayerNorm(dim)
        self.attn2 = A(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=heads, dim_head=dim_head)
        self.norm3 = LayerNorm(dim)
        self.ff = FeedForward(dim)

    def run(self, ctx, x, condition_on_first_frame):
        h = ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        if self.is_temporal:
            x = self.attn1.processor.run(self.attn1, ctx, h, x, adjacent=self.augment_temporal_attention)
        else:
            x = self.attn1.processor.run(self.attn1, ctx, h, x, first_frame=condition_on_first_frame)
        h = ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        kv = self.attn2.text_kv(ctx)
        if self.is_temporal:
            x = self.attn2.processor.run(self.attn2, ctx, h, x, adjacent=False, kv=kv)
        else:
            x = self.attn2.processor.run(self.attn2, ctx, h, x, first_frame=False, kv=kv)
        h = ops.layernorm(x, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        return self.ff.run(h, x)


class Transformer2DConditionModel(nn.Module):
    """``videoldm_transformer_blocks.py:26-318`` (continuous input): GroupNorm(eps 1e-6) -> proj_in -> blocks -> proj_out ->
    + input; the temporal variant blends with ``alpha``.  ``use_linear_projection`` only decides the shape of the proj weights
    (Linear [C, C] or 1 x 1 Conv2d [C, C, 1, 1]): the arithmetic is the same per-token GEMM."""

    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups, n_frames, is_temporal=False,
                 augment_temporal_attention=False, use_linear_projection=True, num_layers=1):
        super().__init__()
        inner = heads * dim_head
        self.use_linear_projection = use_linear_projection
        self.norm = GroupNorm(groups, in_channels, eps=1e-6)
        mk = (lambda a, b: Linear(a, b)) if use_linear_projection else (lambda a, b: Conv2d(a, b, 1))
        self.proj_in = mk(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            BasicConditionalTransformerBlock(inner, heads, dim_head, cross_attention_dim, n_frames, is_temporal, augment_temporal_attention)
            for _ in range(num_layers)])
        self.proj_out = mk(inner, in_channels)
        self.alpha = nn.Parameter(torch.ones(1), requires_grad=False) if is_temporal else None
        self._w_out = self._b_out = None

    def pack(self):
        w, b = self.proj_out.weight.data.reshape(self.proj_out.weight.shape[0], -1), self.proj_out.bias.data
        if self.alpha is not None:   # alpha x + (1 - alpha)(x + f(x)) = x + (1 - alpha) f(x)
            s = 1.0 - float(self.alpha.detach().float().clamp(0, 1))
            w, b = (w.float() * s).to(torch.float16), (b.float() * s).to(torch.float16)
        self._w_out, self._b_out = w.contiguous(), b.contiguous()

    def run(self, ctx, x, condition_on_first_frame=False):
        HW = ctx.H * ctx.W
        h = ops.groupnorm(x, self.norm.weight, self.norm.bias, ctx.stats, HW, groups=self.norm.num_groups, eps=self.norm.eps)
        h = ops.gemm(h, self.proj_in.weight.reshape(self.proj_in.weight.shape[0], -1), bias=self.proj_in.bias)
        for blk in self.transformer_blocks:
            h = blk.run(ctx, h, condition_on_first_frame)
        return ops.gemm(h, self._w_out, bias=self._b_out, residual=x)


class _Conv3DLayer(Conv3dTemporal):
    """``videoldm_unet_blocks.py:316-328``: Conv3d (3,1,1), padding (1,0,0)."""


class TemporalResnetBlock(nn.Module):
    """``videoldm_unet_blocks.py:225-313`` as the decoder calls it (``conv3d(hidden_states)``: no time embedding):
    GroupNorm -> SiLU -> (3,1,1) conv -> GroupNorm -> SiLU -> (3,1,1) conv, residual, alpha blend."""

    def __init__(self, in_channels, groups=32, eps=1e-6, temb_channels=512):
        super().__init__()
        self.norm1 = GroupNorm(groups, in_channels, eps)
        self.conv1 = _Conv3DLayer(in_channels, in_channels)
        self.time_emb_proj = Linear(temb_channels, in_channels)   # in the reference's state dict; unused (temb is None)
        self.norm2 = GroupNorm(groups, in_channels, eps)
        self.conv2 = _Conv3DLayer(in_channels, in_channels)
        self.nonlinearity = SiLU()
        self.alpha = nn.Parameter(torch.ones(1), requires_grad=False)
        self.output_scale_factor = 1.0
        self._w2 = self._b2 = None

    def pack(self):
        self.conv1.pack()
        self.conv2.pack()
        s = 1.0 - float(self.alpha.detach().float().clamp(0, 1))
        self._w2 = (self.conv2._w.float() * s).to(torch.float16).contiguous()
        self._b2 = (self.conv2.bias.data.float() * s).to(torch.float16).contiguous()

    def run(self, ctx, x):
        HW = ctx.H * ctx.W
        g = self.norm1.num_groups
        h = ops.groupnorm(x, self.norm1.weight, self.norm1.bias, ctx.stats, HW, groups=g, eps=self.norm1.eps, silu=True)
        h = ops.gemm(h, self.conv1._w, bias=self.conv1.bias, mode=MODE_TEMPORAL, temporal=(ctx.F, HW))
        h = ops.groupnorm(h, self.norm2.weight, self.norm2.bias, ctx.stats, HW, groups=g, eps=self.norm2.eps, silu=True)
        return ops.gemm(h, self._w2, bias=self._b2, mode=MODE_TEMPORAL, temporal=(ctx.F, HW), residual=x)


# ------------------------------------------------------------------------------------------------- blocks
class _BlockBase(nn.Module):
    """What every block of the UNet shares: packing, and ``enter`` -- one GEMM for the time-embedding projections of all its ResNets."""

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


def _no_conv2d(mode):
    if mode == "conv2d":
        raise NotImplementedError("first_frame_condition_mode='conv2d' is not built (the released ConsistI2V model uses 'concat')")


# ------------------------------------------------------------------------------------------------- the decoder block
class VideoLDMCrossAttnUpBlock(_BlockBase):
    """``videoldm_unet_blocks.py:548-745`` with the released model's options (temporal layers on, rotary temporal position
    embedding, ``first_frame_condition_mode`` "concat" or "none"; the "conv2d" mode is not built).  Constructor argument names are
    the reference's."""

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 num_attention_heads=1, cross_attention_dim=1280, add_upsample=True, use_linear_projection=False, use_temporal=True,
                 augment_temporal_attention=False, n_frames=8, n_temp_heads=8, first_frame_condition_mode="none", rotary_emb=False,
                 transformer_layers_per_block=1, **unused):
        super().__init__()
        if not use_temporal or not rotary_emb or first_frame_condition_mode == "conv2d":
            raise NotImplementedError("native VideoLDMCrossAttnUpBlock: use_temporal=True, rotary_emb=True, first_frame_condition_mode in "
                                      "('none', 'input_only', 'concat') only")
        self.n_frames, self.n_temp_heads, self.num_attention_heads = n_frames, n_temp_heads, num_attention_heads
        self.first_frame_condition_mode = first_frame_condition_mode
        self.has_cross_attention, self.use_temporal = True, True
        self.groups, self.cross_attention_dim = resnet_groups, cross_attention_dim
        self.resnets, self.attentions = nn.ModuleList(), nn.ModuleList()
        self.conv3ds, self.tempo_attns = nn.ModuleList(), nn.ModuleList()
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            self.resnets.append(ResnetBlock2D(rin + skip, out_channels, temb_channels, resnet_groups, resnet_eps))
            self.attentions.append(Transformer2DConditionModel(num_attention_heads, out_channels // num_attention_heads, out_channels,
                                                               cross_attention_dim, resnet_groups, n_frames,
                                                               use_linear_projection=use_linear_projection,
                                                               num_layers=transformer_layers_per_block))
            self.conv3ds.append(TemporalResnetBlock(out_channels))
            self.tempo_attns.append(Transformer2DConditionModel(n_temp_heads, out_channels // n_temp_heads, out_channels,
                                                                cross_attention_dim, resnet_groups, n_frames, is_temporal=True,
                                                                augment_temporal_attention=augment_temporal_attention,
                                                                use_linear_projection=use_linear_projection,
                                                                num_layers=transformer_layers_per_block))
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None
        self._packed = False
        self._w_temb = self._b_temb = None

    def load_state_dict(self, sd, strict=True, **kw):
        out = super().load_state_dict(sd, strict=strict, **kw)   # (weights are converted to the parameters' fp16; alpha / freqs stay fp32)
        self._packed = False
        return out

    def run(self, ctx, x, skips: List[torch.Tensor], out_hw=None):
        """``out_hw``: the reference's ``upsample_size`` (``videoldm_unet_blocks.py:703,744``): the size of the next block's skip connections."""
        cond = self.first_frame_condition_mode not in ("none", "input_only")
        for resnet, conv3d, attn, tattn in zip(self.resnets, self.conv3ds, self.attentions, self.tempo_attns):
            x = resnet.run(ctx, x, skips.pop(), ctx.H, ctx.W)
            x = conv3d.run(ctx, x)
            x = attn.run(ctx, x, condition_on_first_frame=cond)
            x = tattn.run(ctx, x)
        if self.upsamplers is not None:
            x, Ho, Wo = upsample_tokens(self.upsamplers[0].conv, x, ctx.H, ctx.W, out_hw)
            ctx.set_hw(Ho, Wo)
        return x

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None, upsample_size=None, **unused):
        """The reference's call (``videoldm_unet_blocks.py:696-745``): ``hidden_states`` [(b f), C, H, W], the skip tensors of the
        encoder (last one is consumed first), ``temb`` [(b f), D] and ``encoder_hidden_states`` [(b f), L, D] -- both repeated per
        frame by the reference's UNet; the first frame's copy of each batch element is read here."""
        if not self._packed:
            self.pack()
        N, C, H, W = hidden_states.shape
        F = self.n_frames
        B = N // F
        dev = hidden_states.device
        ctx = _Ctx(B, F, H, W, dev, self.groups)

        def tok(t):
            return t.to(torch.float16).permute(0, 2, 3, 1).reshape(N * H * W, t.shape[1]).contiguous()
        emb = temb.to(torch.float16).view(B, F, -1)[:, 0].contiguous()
        ctx.emb = emb
        self.enter(ctx)
        ehs = encoder_hidden_states.to(torch.float16)[::F]
        ctx.L = ehs.shape[1]
        ctx.context = ehs.reshape(B * ctx.L, -1).contiguous()
        y = self.run(ctx, tok(hidden_states), [tok(s) for s in res_hidden_states_tuple],
                     out_hw=None if upsample_size is None else tuple(int(v) for v in upsample_size[-2:]))
        return y.view(N, ctx.H, ctx.W, -1).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------- the other blocks of the UNet
class VideoLDMCrossAttnDownBlock(_BlockBase):
    """``videoldm_unet_blocks.py:343-545``: per layer ResnetBlock2D -> TemporalResnetBlock -> spatial transformer (+ first frame) ->
    temporal transformer; stride-2 ``Downsample2D`` behind the last layer.  Every layer's output is a skip."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, num_attention_heads=1,
                 cross_attention_dim=1280, add_downsample=True, use_linear_projection=False, use_temporal=True,
                 augment_temporal_attention=False, n_frames=8, n_temp_heads=8, first_frame_condition_mode="none", rotary_emb=False,
                 transformer_layers_per_block=1, **unused):
        super().__init__()
        _no_conv2d(first_frame_condition_mode)
        assert use_temporal and rotary_emb
        self.first_frame_condition_mode, self.has_cross_attention, self.n_frames = first_frame_condition_mode, True, n_frames
        self.resnets, self.attentions = nn.ModuleList(), nn.ModuleList()
        self.conv3ds, self.tempo_attns = nn.ModuleList(), nn.ModuleList()
        for i in range(num_layers):
            self.resnets.append(ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_groups, resnet_eps))
            self.attentions.append(Transformer2DConditionModel(num_attention_heads, out_channels // num_attention_heads, out_channels,
                                                               cross_attention_dim, resnet_groups, n_frames,
                                                               use_linear_projection=use_linear_projection,
                                                               num_layers=transformer_layers_per_block))
            self.conv3ds.append(TemporalResnetBlock(out_channels))
            self.tempo_attns.append(Transformer2DConditionModel(n_temp_heads, out_channels // n_temp_heads, out_channels,
                                                                cross_attention_dim, resnet_groups, n_frames, is_temporal=True,
                                                                augment_temporal_attention=augment_temporal_attention,
                                                                use_linear_projection=use_linear_projection,
                                                                num_layers=transformer_layers_per_block))
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def run(self, ctx, x):
        cond = self.first_frame_condition_mode not in ("none", "input_only")
        outs = []
        for resnet, conv3d, attn, tattn in zip(self.resnets, self.conv3ds, self.attentions, self.tempo_attns):
            x = resnet.run(ctx, x, None, ctx.H, ctx.W)
            x = conv3d.run(ctx, x)
            x = attn.run(ctx, x, condition_on_first_frame=cond)
            x = tattn.run(ctx, x)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].conv.tokens(x, ctx.H, ctx.W)
            ctx.set_hw((ctx.H - 1) // 2 + 1, (ctx.W - 1) // 2 + 1)     # (stride 2, padding 1: odd sizes round up)
            outs.append(x)
        return x, outs


class VideoLDMDownBlock(_BlockBase):
    """``videoldm_unet_blocks.py:947-1052`` (diffusers ``DownBlock2D`` + a TemporalResnetBlock per layer; no attention)."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, add_downsample=True,
                 use_temporal=True, n_frames=8, first_frame_condition_mode="none", **unused):
        super().__init__()
        _no_conv2d(first_frame_condition_mode)
        assert use_temporal
        self.has_cross_attention, self.n_frames = False, n_frames
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_groups,
                                                    resnet_eps) for i in range(num_layers)])
        self.conv3ds = nn.ModuleList([TemporalResnetBlock(out_channels) for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def run(self, ctx, x):
        outs = []
        for resnet, conv3d in zip(self.resnets, self.conv3ds):
            x = conv3d.run(ctx, resnet.run(ctx, x, None, ctx.H, ctx.W))
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].conv.tokens(x, ctx.H, ctx.W)
            ctx.set_hw((ctx.H - 1) // 2 + 1, (ctx.W - 1) // 2 + 1)     # (stride 2, padding 1: odd sizes round up)
            outs.append(x)
        return x, outs


class VideoLDMUNetMidBlock2DCrossAttn(_BlockBase):
    """``videoldm_unet_blocks.py:748-945``: resnet, temporal resnet, then per layer spatial transformer -> resnet -> temporal resnet
    (no temporal transformer in the middle)."""

    def __init__(self, in_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, num_attention_heads=1,
                 cross_attention_dim=1280, use_linear_projection=False, use_temporal=True, n_frames=8, first_frame_condition_mode="none",
                 transformer_layers_per_block=1, **unused):
        super().__init__()
        _no_conv2d(first_frame_condition_mode)
        assert use_temporal
        self.first_frame_condition_mode, self.has_cross_attention, self.n_frames = first_frame_condition_mode, True, n_frames
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels, in_channels, temb_channels, resnet_groups, resnet_eps)
                                      for _ in range(num_layers + 1)])
        self.conv3ds = nn.ModuleList([TemporalResnetBlock(in_channels) for _ in range(num_layers + 1)])
        self.attentions = nn.ModuleList([Transformer2DConditionModel(num_attention_heads, in_channels // num_attention_heads, in_channels,
                                                                     cross_attention_dim, resnet_groups, n_frames,
                                                                     use_linear_projection=use_linear_projection,
                                                                     num_layers=transformer_layers_per_block) for _ in range(num_layers)])

    def run(self, ctx, x):
        cond = self.first_frame_condition_mode not in ("none", "input_only")
        x = self.conv3ds[0].run(ctx, self.resnets[0].run(ctx, x, None, ctx.H, ctx.W))
        for attn, resnet, conv3d in zip(self.attentions, self.resnets[1:], self.conv3ds[1:]):
            x = attn.run(ctx, x, condition_on_first_frame=cond)
            x = conv3d.run(ctx, resnet.run(ctx, x, None, ctx.H, ctx.W))
        return x


class VideoLDMUpBlock(_BlockBase):
    """``videoldm_unet_blocks.py:1054-1158`` (diffusers ``UpBlock2D`` + a TemporalResnetBlock per layer; no attention)."""

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 add_upsample=True, use_temporal=True, n_frames=8, first_frame_condition_mode="none", **unused):
        super().__init__()
        _no_conv2d(first_frame_condition_mode)
        assert use_temporal
        self.has_cross_attention, self.n_frames = False, n_frames
        self.resnets = nn.ModuleList()
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            self.resnets.append(ResnetBlock2D(rin + skip, out_channels, temb_channels, resnet_groups, resnet_eps))
        self.conv3ds = nn.ModuleList([TemporalResnetBlock(out_channels) for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def run(self, ctx, x, skips: List[torch.Tensor], out_hw=None):
        for resnet, conv3d in zip(self.resnets, self.conv3ds):
            x = conv3d.run(ctx, resnet.run(ctx, x, skips.pop(), ctx.H, ctx.W))
        if self.upsamplers is not None:
            x, Ho, Wo = upsample_tokens(self.upsamplers[0].conv, x, ctx.H, ctx.W, out_hw)
            ctx.set_hw(Ho, Wo)
        return x


# ------------------------------------------------------------------------------------------------- the UNet
class _Cfg(dict):
    __getattr__ = dict.__getitem__


class TimestepEmbedding(nn.Module):
    """diffusers ``TimestepEmbedding(in, dim, act_fn="silu")``: Linear -> SiLU -> Linear (``videoldm_unet.py:226-246``)."""

    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = Linear(in_channels, time_embed_dim)
        self.linear_2 = Linear(time_embed_dim, time_embed_dim)

    def run(self, x):
        return ops.gemm(ops.gemm(x, self.linear_1.weight, bias=self.linear_1.bias, act=ops.ACT_SILU), self.linear_2.weight,
                        bias=self.linear_2.bias)


class VideoLDMUNet3DConditionModel(nn.Module):
    """``consisti2v/consisti2v/models/videoldm_unet.py:68-1064`` on the HIP kernels, for the configuration family of the released
    ConsistI2V model: temporal layers on, rotary temporal position embedding, ``first_frame_condition_mode`` "concat" (or "none"),
    optional frame-stride conditioning, positional time embedding, no class / addition embeddings.  Module tree and state-dict
    keys are the reference's (``tests/test_consisti2v.py`` loads the reference model's own state dict strictly).

    ``forward(sample [B,C,F',h,w], timestep, encoder_hidden_states [B,L,D], first_frame_latents [B,C,1,h,w], frame_stride)`` as in
    the reference (``:687-1026``): with first-frame conditioning the clean first-frame latent is prepended as frame 0 (F = F' + 1 =
    ``n_frames``), every spatial self-attention also attends to frame 0, every augmented temporal self-attention to its 3 x 3
    neighbourhood in frame 0, and frame 0 of the prediction is dropped."""

    def __init__(self, sample_size=None, in_channels=4, out_channels=4, flip_sin_to_cos=True, freq_shift=0,
                 down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 mid_block_type="UNetMidBlock2DCrossAttn",
                 up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=1280,
                 transformer_layers_per_block=1, attention_head_dim=8, use_linear_projection=False, use_temporal=True, n_frames=8,
                 n_temp_heads=8, first_frame_condition_mode="none", augment_temporal_attention=False, temp_pos_embedding="sinusoidal",
                 use_frame_stride_condition=False, **unused):
        super().__init__()
        if not use_temporal or temp_pos_embedding != "rotary" or not flip_sin_to_cos or freq_shift != 0 or mid_block_type != "UNetMidBlock2DCrossAttn":
            raise NotImplementedError("native VideoLDMUNet3DConditionModel: use_temporal=True, temp_pos_embedding='rotary', flip_sin_to_cos=True, "
                                      "freq_shift=0, mid_block_type='UNetMidBlock2DCrossAttn' only")
        _no_conv2d(first_frame_condition_mode)
        nb = len(block_out_channels)
        tup = lambda v: tuple(v) if isinstance(v, (list, tuple)) else (v,) * nb
        heads, lpb, tlpb, cad = tup(attention_head_dim), tup(layers_per_block), tup(transformer_layers_per_block), tup(cross_attention_dim)
        self.config = _Cfg(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels, n_frames=n_frames,
                           first_frame_condition_mode=first_frame_condition_mode, cross_attention_dim=cross_attention_dim,
                           block_out_channels=tuple(block_out_channels), use_frame_stride_condition=use_frame_stride_condition)
        boc = block_out_channels
        ted = boc[0] * 4
        self.groups = norm_num_groups
        self.conv_in = Conv2d(in_channels, boc[0], 3, padding=1, pad_cin_to=64)
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.use_frame_stride_condition = use_frame_stride_condition
        if use_frame_stride_condition:
            self.frame_stride_embedding = TimestepEmbedding(boc[0], ted)
        common = dict(temb_channels=ted, resnet_eps=norm_eps, resnet_groups=norm_num_groups, use_linear_projection=use_linear_projection,
                      use_temporal=True, augment_temporal_attention=augment_temporal_attention, n_frames=n_frames, n_temp_heads=n_temp_heads,
                      first_frame_condition_mode=first_frame_condition_mode, rotary_emb=True)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, typ in enumerate(down_block_types):
            cin, out = out, boc[i]
            kw = dict(in_channels=cin, out_channels=out, num_layers=lpb[i], add_downsample=i != nb - 1, **common)
            if typ == "CrossAttnDownBlock2D":
                self.down_blocks.append(VideoLDMCrossAttnDownBlock(num_attention_heads=heads[i], cross_attention_dim=cad[i],
                                                                   transformer_layers_per_block=tlpb[i], **kw))
            elif typ == "DownBlock2D":
                self.down_blocks.append(VideoLDMDownBlock(**kw))
            else:
                raise NotImplementedError(typ)
        self.mid_block = VideoLDMUNetMidBlock2DCrossAttn(in_channels=boc[-1], num_attention_heads=heads[-1], cross_attention_dim=cad[-1],
                                                         transformer_layers_per_block=tlpb[-1], **common)
        self.up_blocks = nn.ModuleList()
        rboc, rheads, rlpb, rtlpb, rcad = boc[::-1], heads[::-1], lpb[::-1], tlpb[::-1], cad[::-1]
        out = rboc[0]
        for i, typ in enumerate(up_block_types):
            prev, out = out, rboc[i]
            cin = rboc[min(i + 1, nb - 1)]
            kw = dict(in_channels=cin, out_channels=out, prev_output_channel=prev, num_layers=rlpb[i] + 1, add_upsample=i != nb - 1, **common)
            if typ == "CrossAttnUpBlock2D":
                self.up_blocks.append(VideoLDMCrossAttnUpBlock(num_attention_heads=rheads[i], cross_attention_dim=rcad[i],
                                                               transformer_layers_per_block=rtlpb[i], **kw))
            elif typ == "UpBlock2D":
                self.up_blocks.append(VideoLDMUpBlock(**kw))
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
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1   # captured graphs point at the packed tensors of one generation

    def load_state_dict(self, sd, strict=True, **kw):
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._packed = False
        return out

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states=None, first_frame_latents=None, frame_stride=None, return_dict=True, **unused):
        if not self._packed:
            self.pack()
        dev = sample.device
        cfgd = self.config
        if cfgd.first_frame_condition_mode != "none":
            assert first_frame_latents is not None
            sample = torch.cat([first_frame_latents.to(sample.dtype), sample], dim=2)
        B, C, F, H, W = sample.shape
        ctx = _Ctx(B, F, H, W, dev, self.groups)
        c0 = cfgd.block_out_channels[0]
        # (device tensors pass through untouched: a captured step keeps the timestep in a static buffer)
        t = torch.as_tensor(timestep, device=dev).reshape(-1).float().expand(B).contiguous()
        emb = self.time_embedding.run(ops.timestep_embedding(t, c0))
        if self.use_frame_stride_condition:
            fs = torch.as_tensor(frame_stride, device=dev).reshape(-1).float().expand(B).contiguous()
            emb = ops.add(emb, self.frame_stride_embedding.run(ops.timestep_embedding(fs, c0)))
        ctx.emb = emb
        ehs = encoder_hidden_states.to(torch.float16)
        ctx.L = ehs.shape[1]
        ctx.context = ehs.reshape(B * ctx.L, -1).contiguous()
        xin = torch.zeros((B * F * H * W, 64), dtype=torch.float16, device=dev)
        ops.ncfhw_to_tokens(sample.to(torch.float16).contiguous(), xin, col0=0)
        x = self.conv_in.tokens(xin, H, W)
        skips = [x]
        sizes = [(H, W)]     # per resolution level: the up path returns to exactly these (``upsample_size``, ``videoldm_unet.py:726-734,990-1010``)
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
        x = ops.groupnorm(x, self.conv_norm_out.weight, self.conv_norm_out.bias, ctx.stats, H * W, groups=self.groups,
                          eps=self.conv_norm_out.eps, silu=True)
        vtok = torch.empty((x.shape[0], 8), dtype=torch.float16, device=dev)
        self.conv_out.tokens(x, H, W, out=vtok)
        out = ops.tokens_to_ncfhw(vtok, B, cfgd.out_channels, F, H, W)
        if cfgd.first_frame_condition_mode != "none":
            out = out[:, :, 1:]
        if not return_dict:
            return (out,)
        return types.SimpleNamespace(sample=out)


# ------------------------------------------------------------------------------------------------- hook registration
_UP_RES = {1: [0, 1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}      # consisti2v/pnp_utils.py:22 (register_time)
_INJ_RES = {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}        # :228, :356 (decoder blocks 4-11)


def _sched(injection_schedule):
    """Schedules may be tensors / lists in the reference's scripts; membership tests must not sync the device."""
    if injection_schedule is None:
        return None
    if isinstance(injection_schedule, torch.Tensor):
        return frozenset(int(v) for v in injection_schedule.tolist())
    return frozenset(int(v) for v in injection_schedule)


def register_time(model, t):
    """``consisti2v/pnp_utils.py:19-29``."""
    t = int(t)
    setattr(model.unet.up_blocks[1].resnets[1], "t", t)
    for res, blocks in _UP_RES.items():
        for block in blocks:
            setattr(model.unet.up_blocks[res].attentions[block].transformer_blocks[0].attn1.processor, "t", t)
            setattr(model.unet.up_blocks[res].tempo_attns[block].transformer_blocks[0].attn1.processor, "t", t)


def injection_state(model):
    """(conv, spatial, temporal) injection on / off at the registered timestep: which launches a forward issues -- the key of a
    captured graph."""
    r = model.unet.up_blocks[1].resnets[1]
    a = model.unet.up_blocks[2].attentions[0].transformer_blocks[0].attn1.processor
    m = model.unet.up_blocks[2].tempo_attns[0].transformer_blocks[0].attn1.processor
    return (pnp_on(r.t, r.injection_schedule), pnp_on(a.t, getattr(a, "injection_schedule", None)),
            pnp_on(m.t, getattr(m, "injection_schedule", None)))


def clear_time(model):
    """No hook fires until the next ``register_time`` (inversion and plain sampling run on a hook-free UNet in the reference: its
    two stages are separate processes)."""
    model.unet.up_blocks[1].resnets[1].t = None
    for res, blocks in _UP_RES.items():
        for block in blocks:
            model.unet.up_blocks[res].attentions[block].transformer_blocks[0].attn1.processor.t = None
            model.unet.up_blocks[res].tempo_attns[block].transformer_blocks[0].attn1.processor.t = None


def register_conv_injection(model, injection_schedule):
    """``consisti2v/pnp_utils.py:39-128``: the native ``ResnetBlock2D.run`` holds the injection (source branch's conv features
    into the other two branches); only the schedule is attached."""
    setattr(model.unet.up_blocks[1].resnets[1], "injection_schedule", _sched(injection_schedule))


def register_spatial_attention_pnp(model, injection_schedule):
    """``consisti2v/pnp_utils.py:131-240``."""
    for res, blocks in _INJ_RES.items():
        for block in blocks:
            module = model.unet.up_blocks[res].attentions[block].transformer_blocks[0].attn1
            module.processor = HipSpaAttnProcessor(_sched(injection_schedule))


def register_temp_attention_pnp(model, injection_schedule):
    """``consisti2v/pnp_utils.py:244-362``."""
    for res, blocks in _INJ_RES.items():
        for block in blocks:
            module = model.unet.up_blocks[res].tempo_attns[block].transformer_blocks[0].attn1
            module.processor = HipTmpAttnProcessor(_sched(injection_schedule))
