"""Thin torch-tensor -> C-ABI wrappers.  Torch is plumbing only (device memory + current stream)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import AttnDesc, GemmDesc

MODE_LINEAR, MODE_CONV2D, MODE_TEMPORAL = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GELU, ACT_GEGLU, ACT_F32OUT = 0, 1, 2, 3, 4

# global knobs (tests flip them to cross-check kernel variants)
FORCE_NAIVE = False   # route GEMM / attention through the reference-grade kernels
ATTN_FLAGS = int(os.environ.get("ANYV2V_ATTN_FLAGS", "0"))  # bit1 (2): no short kernel; bit3 (8): PnP launches as per-branch aliasing (no shared-softmax kernel)
GEMM_FLAGS = int(os.environ.get("ANYV2V_GEMM_FLAGS", "0"))  # bit2 (4): no persistent 192x320 kernel; bit3 (8): force it; bit4 (16): no split-K
USE_GLDS = os.environ.get("ANYV2V_GLDS", "1") == "1"   # LDS-DMA (global_load_lds) staging variant of the GEMM


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _rowmajor(t: torch.Tensor, name: str):
    assert t.dim() == 2 and t.stride(1) == 1, f"{name}: need a 2-D row-major (stride(1)==1) fp16 view, got {tuple(t.shape)} {t.stride()}"
    assert t.dtype == torch.float16, f"{name}: fp16 expected, got {t.dtype}"
    assert t.is_cuda, f"{name}: device tensor expected"


_WS = {}
WS_SLOT = 0   # callers that enqueue on several streams at once (two step engines side by side) give each stream its own scratch


class workspace_slot:
    """``with ops.workspace_slot(k):`` -- the launches enqueued inside use scratch buffer ``k`` (re-entrant, host-side)."""

    def __init__(self, slot: int):
        self.slot, self.prev = int(slot), 0

    def __enter__(self):
        global WS_SLOT
        self.prev, WS_SLOT = WS_SLOT, self.slot
        return self

    def __exit__(self, *exc):
        global WS_SLOT
        WS_SLOT = self.prev
        return False


def _workspace(device) -> torch.Tensor:
    """Persistent fp32 scratch for split-K partial tiles and stream-K slabs (128 MiB per device and slot -- the split-K rules keep
    planning against 64 MiB, the stream-K form needs 126 MB; allocated once, outside any graph capture because the first GEMM of a
    process always runs eagerly during warm-up)."""
    key = f"{device}:{WS_SLOT}"
    if key not in _WS:
        _WS[key] = torch.empty(32 * 1024 * 1024, dtype=torch.float32, device=device)
    return _WS[key]


def gemm(a0: torch.Tensor, w: torch.Tensor, *, a1: Optional[torch.Tensor] = None, bias=None, rowvec=None,
         rowvec_div: int = 0, residual=None, act: int = ACT_NONE, out: Optional[torch.Tensor] = None,
         mode: int = MODE_LINEAR, conv=None, temporal=None, M: Optional[int] = None, naive: bool = False, ln=None):
    """out[M, N'] = epilogue(gather(A) @ W^T).  ``w`` is [N, taps*K] packed (see anyv2v_hip.h).

    ``ln`` = (c1 fp32 [N], eps): LayerNorm folded into the projection -- ``a0`` holds the un-normalised rows, ``w`` / ``bias`` the
    gamma- / beta-folded weights (``ln_fold``); only shapes ``ln_gemm_supported`` accepts.

    conv = (Hi, Wi, Ho, Wo, stride, up[, asym]) for MODE_CONV2D; temporal = (F, HW) for MODE_TEMPORAL.
    ``M`` overrides the row count (output rows); A may have a different number of rows for convs.
    """
    lib = _lib.load()
    _rowmajor(a0, "A0")
    _rowmajor(w, "W")
    C0 = a0.shape[1]
    C1 = 0
    if a1 is not None:
        _rowmajor(a1, "A1")
        C1 = a1.shape[1]
    taps = 1 if mode == MODE_LINEAR else (9 if mode == MODE_CONV2D else 3)
    N = w.shape[0]
    assert w.shape[1] == taps * (C0 + C1), f"W is {tuple(w.shape)}, expected [{N}, {taps}*({C0}+{C1})]"
    assert w.stride(0) == w.shape[1], "W must be contiguous"
    if M is None:
        M = a0.shape[0]
    n_out = N // 2 if act == ACT_GEGLU else N
    if act == ACT_F32OUT:  # raw fp32 accumulator (+bias) -> float32 matrix
        if out is None:
            out = torch.empty((M, n_out), dtype=torch.float32, device=a0.device)
        assert out.dim() == 2 and out.stride(1) == 1 and out.dtype == torch.float32 and out.is_cuda
        assert rowvec is None and residual is None
    else:
        if out is None:
            out = torch.empty((M, n_out), dtype=torch.float16, device=a0.device)
        _rowmajor(out, "C")
    assert out.shape[0] >= M and out.shape[1] >= n_out
    d = GemmDesc()
    d.A0, d.A1, d.W, d.C = _p(a0), _p(a1), _p(w), _p(out)
    d.bias, d.rowvec, d.R = _p(bias), _p(rowvec), _p(residual)
    d.M, d.N, d.C0, d.C1 = M, N, C0, C1
    d.lda0 = a0.stride(0)
    d.lda1 = a1.stride(0) if a1 is not None else 0
    d.ldc = out.stride(0)
    d.ldr = residual.stride(0) if residual is not None else 0
    d.ldrv = rowvec.stride(0) if rowvec is not None else 0
    d.rowvec_div = rowvec_div
    d.mode = mode
    if mode == MODE_CONV2D:
        d.Hi, d.Wi, d.Ho, d.Wo, d.stride, d.up = conv[:6]
        d.asym = conv[6] if len(conv) > 6 else 0  # 1: pad only right / bottom (AutoencoderKL Downsample2D)
    elif mode == MODE_TEMPORAL:
        d.F, d.HW = temporal
    d.act = act
    d.flags = (1 if (naive or FORCE_NAIVE) else 0) | (2 if USE_GLDS else 0) | GEMM_FLAGS
    if ln is not None:
        c1, eps = ln
        assert c1.dtype == torch.float32 and c1.is_cuda and c1.numel() >= N and c1.is_contiguous()
        d.ln_c1, d.ln_eps = _p(c1), float(eps)
    ws = _workspace(a0.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    _lib.check(lib.anyv2v_gemm_f16(C.byref(d), _stream()), "anyv2v_gemm_f16")
    return out


FF_FUSED = os.environ.get("ANYV2V_FF_FUSED", "1") == "1"   # A/B switch: fused feed-forward kernel (ff_fused.hip) vs GEGLU GEMM + Linear


def ff_fused_supported(M: int, C: int, H: int) -> bool:
    """Shapes the fused feed-forward kernel covers AND pays on: the 320-channel transformer blocks at the 64x64 level's row counts
    (same threshold, on hinted rows, as the weight-stationary GEMMs it replaces)."""
    return FF_FUSED and not FORCE_NAIVE and USE_GLDS and C == 320 and H == 1280 and M * _HINT[0] // _HINT[1] >= 32768


def ff_pack_w2(w2: torch.Tensor) -> torch.Tensor:
    """``Linear(H, C).weight`` [C, H] -> the fused kernel's layout [H / 32][C][32]: slab-major, and inside a slab of 32 hidden units
    column 8 q + e holds hidden unit 4 q + e (e < 4) or 16 + 4 q + (e - 4) (e >= 4) -- the MFMA K-slot order in which the kernel's
    GEGLU registers already hold the hidden values (include/anyv2v_hip.h, AnyV2VFFDesc)."""
    C, H = w2.shape
    assert H % 32 == 0
    q, e = torch.arange(4).view(4, 1), torch.arange(8).view(1, 8)
    perm = torch.where(e < 4, 4 * q + e, 16 + 4 * q + (e - 4)).reshape(32).to(w2.device)   # slot -> hidden unit of the slab
    return w2.view(C, H // 32, 32)[:, :, perm].permute(1, 0, 2).contiguous()


def ff_geglu(x: torch.Tensor, w1p: torch.Tensor, b1p: torch.Tensor, w2s: torch.Tensor, b2: torch.Tensor, residual=None, out=None):
    """y = GEGLU(x W1^T + b1) W2^T + b2 (+ residual) in one kernel (``anyv2v_ff_geglu_f16``); ``w1p`` / ``b1p`` as ``GEGLU.pack``
    interleaves them, ``w2s`` from ``ff_pack_w2``."""
    lib = _lib.load()
    M, Cc = x.shape
    H = w2s.shape[0] * 32
    assert x.dtype == torch.float16 and x.stride(1) == 1 and w1p.is_contiguous() and w2s.is_contiguous()
    assert tuple(w1p.shape) == (2 * H, Cc) and tuple(w2s.shape) == (H // 32, Cc, 32) and b1p.numel() == 2 * H and b2.numel() == Cc
    if out is None:
        out = torch.empty((M, Cc), dtype=torch.float16, device=x.device)
    for name, t in (("residual", residual), ("out", out)):   # the kernel reads / writes them as 8-byte vectors at row * ld + column
        assert t is None or (tuple(t.shape) == (M, Cc) and t.dtype == torch.float16 and t.stride(1) == 1 and t.device == x.device), \
            f"ff_geglu: {name} must be an fp16 [{M}, {Cc}] matrix with unit column stride on {x.device}"
    d = _lib.FFDesc()
    d.X, d.W1, d.b1, d.W2, d.b2, d.Y = _p(x), _p(w1p), _p(b1p), _p(w2s), _p(b2), _p(out)
    d.R = _p(residual) if residual is not None else None
    d.M, d.C, d.H = M, Cc, H
    d.ldx, d.ldy, d.ldr = x.stride(0), out.stride(0), (residual.stride(0) if residual is not None else 0)
    d.flags = d.reserved0 = 0
    _lib.check(lib.anyv2v_ff_geglu_f16(C.byref(d), _stream()), "anyv2v_ff_geglu_f16")
    return out


def ln_gemm_supported(M: int, K: int, N: int, act: int = ACT_NONE, hinted: bool = True) -> bool:
    """Shapes for which ``gemm(..., ln=...)`` runs (the weight-stationary kernel, gemm_ws.hip) AND pays: the 64x64 level's row
    counts.  Mirrors the library's own check; everything else runs ``layernorm`` + ``gemm``.  ``hinted=False``: ``M`` is already a
    canonical row count (one branch's rows) and is compared as it is, whatever batch hint is in force."""
    if FORCE_NAIVE or not USE_GLDS or (GEMM_FLAGS & (512 | 4)) or (M * _HINT[0] // _HINT[1] if hinted else M) < 32768:
        return False
    if K == 320:
        return N % 160 == 0 and N // 160 <= 32
    return K == 512 and act == ACT_GEGLU and N % 128 == 0 and N // 128 <= 32


def ln_fold(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """(W', b', c1) of LN(x) W^T + b = rstd (x W'^T - mean c1) + b':  W' = W diag(gamma) rounded to fp16, c1 = row sums of that
    ROUNDED W' in fp32 (so that x W'^T - mean c1 is exactly the product of the centred row with the weights the kernel multiplies
    by: no cancellation error from the rounding), b' = b + W beta."""
    wf = w.float() * gamma.float()[None, :]
    w16 = wf.to(torch.float16).contiguous()
    c1 = w16.float().sum(1).contiguous()
    b = w.float() @ beta.float()
    if bias is not None:
        b = b + bias.float()
    return w16, b.to(torch.float16).contiguous(), c1


def gn_scratch_floats(M: int, rows_per_group: int, groups: int = 32) -> int:
    return (M // rows_per_group) * groups * 2 * 257  # == anyv2v_groupnorm_scratch_floats (1 + 256 chunks)


def groupnorm(x0: torch.Tensor, gamma, beta, stats: torch.Tensor, rows_per_group: int, *, x1=None, groups: int = 32,
              eps: float = 1e-5, silu: bool = False, out=None, shard=None):
    """``shard`` = (shards, all_reduce_sum): this rank holds 1/shards of every statistics group (frame- / pixel-parallel
    clip, ``anyv2v_amd.parallel.FrameParallel``); the partial sums are added over the ranks between the two kernels."""
    if shard is not None:
        return _groupnorm_sharded(x0, gamma, beta, stats, rows_per_group, x1, groups, eps, silu, out, shard)
    lib = _lib.load()
    _rowmajor(x0, "X0")
    assert x0.is_contiguous()
    M, C0 = x0.shape
    C1 = 0
    if x1 is not None:
        _rowmajor(x1, "X1")
        assert x1.is_contiguous() and x1.shape[0] == M
        C1 = x1.shape[1]
    if out is None:
        out = torch.empty((M, C0 + C1), dtype=torch.float16, device=x0.device)
    assert out.is_contiguous()
    need = gn_scratch_floats(M, rows_per_group, groups)
    assert stats.dtype == torch.float32 and stats.numel() >= need, "GroupNorm scratch too small"
    _lib.check(lib.anyv2v_groupnorm_f16(_p(x0), _p(x1), C0, C1, _p(out), _p(gamma), _p(beta), _p(stats), M,
                                        rows_per_group, groups, eps, int(silu), _stream()), "anyv2v_groupnorm_f16")
    return out


def _groupnorm_sharded(x0, gamma, beta, stats, rows_per_group, x1, groups, eps, silu, out, shard):
    lib = _lib.load()
    shards, all_reduce_sum = shard
    _rowmajor(x0, "X0")
    assert x0.is_contiguous()
    M, C0 = x0.shape
    C1 = 0
    if x1 is not None:
        _rowmajor(x1, "X1")
        assert x1.is_contiguous() and x1.shape[0] == M
        C1 = x1.shape[1]
    if out is None:
        out = torch.empty((M, C0 + C1), dtype=torch.float16, device=x0.device)
    assert out.is_contiguous()
    n = int(lib.anyv2v_groupnorm_partial_floats(M, rows_per_group, groups, C0 + C1))
    assert n > 0 and stats.dtype == torch.float32 and stats.numel() >= n, "GroupNorm scratch too small"
    _lib.check(lib.anyv2v_groupnorm_partial_f16(_p(x0), _p(x1), C0, C1, _p(stats), M, rows_per_group, groups, _stream()),
               "anyv2v_groupnorm_partial_f16")
    all_reduce_sum(stats[:n])
    _lib.check(lib.anyv2v_groupnorm_apply_f16(_p(x0), _p(x1), C0, C1, _p(out), _p(gamma), _p(beta), _p(stats), M,
                                              rows_per_group, groups, eps, int(silu), int(shards), _stream()),
               "anyv2v_groupnorm_apply_f16")
    return out


def layernorm(x: torch.Tensor, gamma, beta, eps: float = 1e-5, out=None):
    lib = _lib.load()
    _rowmajor(x, "X")
    assert x.is_contiguous()
    M, Cc = x.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.anyv2v_layernorm_f16(_p(x), _p(out), _p(gamma), _p(beta), M, Cc, eps, _stream()), "anyv2v_layernorm_f16")
    return out


def softmax_rows(s: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None):
    """fp32 logits [rows, cols] -> fp16 softmax(scale * s) (row-wise)."""
    lib = _lib.load()
    assert s.dim() == 2 and s.stride(1) == 1 and s.dtype == torch.float32 and s.is_cuda
    rows, cols = s.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float16, device=s.device)
    _rowmajor(out, "P")
    _lib.check(lib.anyv2v_softmax_rows_f32_f16(_p(s), s.stride(0), _p(out), out.stride(0), rows, cols, float(scale), _stream()),
               "anyv2v_softmax_rows_f32_f16")
    return out


def attention(q, k, v, out, *, batch, heads, Sq, Sk, inner=1, q_strides, kv_strides, kv_div=1, qk_mod=0,
              scale=0.125, head_dim=64, naive=False, causal=False, bias=None):
    """Strided multi-head attention over token matrices; see anyv2v_hip.h for the addressing.  ``causal`` (CLIP text tower)
    and head_dim != 64 run on the small generic kernel; ``bias`` (fp32 [heads, Sq, Sk], added to the scaled scores) on the
    generic one."""
    lib = _lib.load()
    for t, n in ((q, "Q"), (k, "K"), (v, "V"), (out, "O")):
        _rowmajor(t, n)
    d = AttnDesc()
    d.Q, d.K, d.V, d.O = _p(q), _p(k), _p(v), _p(out)
    d.ldq, d.ldk, d.ldv, d.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    d.batch, d.heads, d.Sq, d.Sk, d.inner = batch, heads, Sq, Sk, inner
    d.q_outer, d.q_inner, d.q_seq = q_strides
    d.kv_outer, d.kv_inner, d.kv_seq = kv_strides
    d.kv_div, d.qk_mod, d.scale = kv_div, qk_mod, scale
    d.flags = (1 if (naive or FORCE_NAIVE) else 0) | ATTN_FLAGS | (16 if causal else 0)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous() and tuple(bias.shape) == (heads, Sq, Sk) and bias.device == q.device
        _lib.check(lib.anyv2v_attention_bias_f16(C.byref(d), head_dim, _p(bias), _stream()), "anyv2v_attention_bias_f16")
    elif head_dim == 64 and not causal:
        _lib.check(lib.anyv2v_attention_f16(C.byref(d), _stream()), "anyv2v_attention_f16")
    else:
        _lib.check(lib.anyv2v_attention_small_f16(C.byref(d), head_dim, _stream()), "anyv2v_attention_small_f16")
    return out


def silu(x: torch.Tensor, out=None):
    lib = _lib.load()
    assert x.is_contiguous() and x.dtype == torch.float16
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.anyv2v_silu_f16(_p(x), _p(out), x.numel(), _stream()), "anyv2v_silu_f16")
    return out


def add(a: torch.Tensor, b: torch.Tensor, out=None):
    lib = _lib.load()
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    if out is None:
        out = torch.empty_like(a)
    _lib.check(lib.anyv2v_add_f16(_p(a), _p(b), _p(out), a.numel(), _stream()), "anyv2v_add_f16")
    return out


def timestep_embedding(t_f32: torch.Tensor, dim: int, out=None):
    lib = _lib.load()
    assert t_f32.dtype == torch.float32 and t_f32.is_contiguous()
    B = t_f32.numel()
    if out is None:
        out = torch.empty((B, dim), dtype=torch.float16, device=t_f32.device)
    _lib.check(lib.anyv2v_timestep_embedding_f16(_p(t_f32), _p(out), B, dim, _stream()), "anyv2v_timestep_embedding_f16")
    return out


def ncfhw_to_tokens(x: torch.Tensor, out: torch.Tensor, col0: int = 0):
    lib = _lib.load()
    B, Cc, F, H, W = x.shape
    assert x.is_contiguous() and x.dtype == torch.float16
    _rowmajor(out, "Y")
    _lib.check(lib.anyv2v_ncfhw_to_tokens_f16(_p(x), _p(out), B, Cc, F, H * W, out.stride(0), col0, _stream()),
               "anyv2v_ncfhw_to_tokens_f16")
    return out


def tokens_to_ncfhw(x: torch.Tensor, B: int, Cc: int, F: int, H: int, W: int, col0: int = 0, out=None):
    lib = _lib.load()
    _rowmajor(x, "X")
    if out is None:
        out = torch.empty((B, Cc, F, H, W), dtype=torch.float16, device=x.device)
    _lib.check(lib.anyv2v_tokens_to_ncfhw_f16(_p(x), _p(out), B, Cc, F, H * W, x.stride(0), col0, _stream()),
               "anyv2v_tokens_to_ncfhw_f16")
    return out


def adaptive_avgpool(x: torch.Tensor, N: int, Hi: int, Wi: int, Ho: int, Wo: int):
    lib = _lib.load()
    _rowmajor(x, "X")
    assert x.is_contiguous()
    Cc = x.shape[1]
    out = torch.empty((N * Ho * Wo, Cc), dtype=torch.float16, device=x.device)
    _lib.check(lib.anyv2v_adaptive_avgpool_f16(_p(x), _p(out), N, Hi, Wi, Ho, Wo, Cc, _stream()), "anyv2v_adaptive_avgpool_f16")
    return out


def copy_cols(x: torch.Tensor, xcol0: int, y: torch.Tensor, ycol0: int, ncols: int):
    lib = _lib.load()
    _rowmajor(x, "X")
    _rowmajor(y, "Y")
    assert x.shape[0] == y.shape[0]
    _lib.check(lib.anyv2v_copy_cols_f16(_p(x), x.stride(0), xcol0, _p(y), y.stride(0), ycol0, x.shape[0], ncols, _stream()),
               "anyv2v_copy_cols_f16")
    return y


def gather_rows(x: torch.Tensor, xcol0: int, idx: torch.Tensor, y: torch.Tensor, ycol0: int, ncols: int):
    """y[m, ycol0:ycol0+ncols] = x[idx[m], xcol0:xcol0+ncols]; ``idx`` int32 on the device, one entry per row of ``y``."""
    lib = _lib.load()
    _rowmajor(x, "X")
    _rowmajor(y, "Y")
    assert idx.dtype == torch.int32 and idx.is_contiguous() and idx.numel() == y.shape[0] and idx.device == x.device
    _lib.check(lib.anyv2v_gather_rows_f16(_p(x), x.stride(0), xcol0, _p(idx), _p(y), y.stride(0), ycol0, y.shape[0], ncols,
                                          _stream()), "anyv2v_gather_rows_f16")
    return y


def rotary(x: torch.Tensor, col0: int, rot_dim: int, rows_per_pos: int, n_pos: int, theta: float = 10000.0, windows: int = 1,
           window_stride: int = 0):
    """In-place rotary position embedding of columns [col0 + w * window_stride, ... + rot_dim), w < windows (interleaved pairs);
    the position of row r is (r // rows_per_pos) % n_pos -- see include/anyv2v_hip.h."""
    lib = _lib.load()
    _rowmajor(x, "X")
    _lib.check(lib.anyv2v_rotary_f16(_p(x), x.stride(0), x.shape[0], col0, rot_dim, windows, window_stride, rows_per_pos, n_pos,
                                     float(theta), _stream()), "anyv2v_rotary_f16")
    return x


def cfg_ddim_step(vtok: torch.Tensor, b_unc: int, b_cond: int, guidance: float, coef: torch.Tensor,
                  lat: torch.Tensor, out: torch.Tensor):
    """lat/out: [1, C, F, H, W] fp16; vtok: [(nb F) HW, ld] channels-last v-prediction; coef: 4 device floats."""
    lib = _lib.load()
    _rowmajor(vtok, "V")
    assert lat.is_contiguous() and out.is_contiguous() and lat.dtype == torch.float16 and lat.shape[0] == 1
    assert coef.dtype == torch.float32 and coef.numel() >= 4
    _, Cc, F, H, W = lat.shape
    _lib.check(lib.anyv2v_cfg_ddim_step_f16(_p(vtok), vtok.stride(0), b_unc, b_cond, float(guidance), _p(coef), _p(lat),
                                            _p(out), Cc, F, H * W, _stream()), "anyv2v_cfg_ddim_step_f16")
    return out


def ddim_step(v: torch.Tensor, x: torch.Tensor, sa_t: float, sb_t: float, sa_p: float, sb_p: float, out=None):
    lib = _lib.load()
    v = v.to(torch.float16).contiguous()
    x = x.to(torch.float16).contiguous()
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.anyv2v_ddim_step_f16(_p(v), _p(x), _p(out), sa_t, sb_t, sa_p, sb_p, x.numel(), _stream()),
               "anyv2v_ddim_step_f16")
    return out


PRED_V, PRED_EPSILON, PRED_SAMPLE = 0, 1, 2


def guided_step(e: torch.Tensor, x: torch.Tensor, coef, *, b_txt: int, b_unc: int = -1, b_img: int = -1, g_txt: float = 1.0,
                g_img: float = 1.0, prediction: int = PRED_V, out=None, noise=None, sigma: float = 0.0):
    """e: [nb, ...] the UNet's prediction of every branch; x: one branch's latents; coef = (sa_t, sb_t, c_x0, c_eps):
    y = c_x0 x0 + c_eps eps (+ sigma noise) -- DDIM: (c_x0, c_eps) = (sa_p, sb_p)."""
    lib = _lib.load()
    assert e.is_contiguous() and x.is_contiguous() and e.dtype == x.dtype == torch.float16
    n = x.numel()
    assert e.numel() % n == 0 and max(b_txt, b_unc, b_img) < e.numel() // n
    if out is None:
        out = torch.empty_like(x)
    sa_t, sb_t, sa_p, sb_p = (float(c) for c in coef)
    if noise is not None or sigma != 0.0:
        assert noise is None or (noise.is_contiguous() and noise.dtype == torch.float16 and noise.numel() == n)
        _lib.check(lib.anyv2v_guided_step_noise_f16(_p(e), n, b_unc, b_img, b_txt, float(g_img), float(g_txt), int(prediction), sa_t, sb_t,
                                                    sa_p, sb_p, _p(x), _p(out), _p(noise), float(sigma), _stream()),
                   "anyv2v_guided_step_noise_f16")
        return out
    _lib.check(lib.anyv2v_guided_step_f16(_p(e), n, b_unc, b_img, b_txt, float(g_img), float(g_txt), int(prediction), sa_t, sb_t, sa_p,
                                          sb_p, _p(x), _p(out), _stream()), "anyv2v_guided_step_f16")
    return out


_HINT = [1, 1]   # mirrored for the host-side decisions (ln_gemm_supported)


def set_batch_hint(num: int, den: int):
    """Switch the hint inside a forward (the shared stem of a CFG batch has one branch less than the layers behind it, so its
    ratio differs: ``I2VGenXLUNet._forward_core``)."""
    _lib.check(_lib.load().anyv2v_set_batch_hint(int(num), int(den)), "anyv2v_set_batch_hint")
    _HINT[:] = [int(num), int(den)]


class batch_hint:
    """``with ops.batch_hint(3, 2): ...`` -- the launches inside choose kernels / split-K factors / GroupNorm chunking as if they had
    3/2 of their rows (``anyv2v_set_batch_hint``): a [negative, editing] step then computes, bit for bit, what the three-branch step
    computes for those branches.  Also valid around a HIP-graph capture (the choices are baked at capture time)."""

    def __init__(self, num: int, den: int):
        self.num, self.den = int(num), int(den)

    def __enter__(self):
        _lib.check(_lib.load().anyv2v_set_batch_hint(self.num, self.den), "anyv2v_set_batch_hint")
        _HINT[:] = [self.num, self.den]
        return self

    def __exit__(self, *exc):
        _lib.check(_lib.load().anyv2v_set_batch_hint(1, 1), "anyv2v_set_batch_hint")
        _HINT[:] = [1, 1]
        return False


def selftest(scratch: torch.Tensor):
    lib = _lib.load()
    assert scratch.dtype == torch.uint8 and scratch.is_contiguous()
    _lib.check(lib.anyv2v_selftest(_p(scratch), scratch.numel(), _stream()), "anyv2v_selftest")
