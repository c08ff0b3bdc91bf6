"""GPU parity checks shared by the pytest files (``-m gpu``) and ``tools/gpu_check.py`` (diagnostic report).

Each check returns a dict(name=..., err=..., tol=..., ok=...).  Kernel-level checks compare the HIP kernels with
plain PyTorch fp32 ops on the same fp16-rounded inputs; model-level checks compare with the CPU oracle
(``oracle/``) and with the golden fixtures generated from the reference's own code (``tests/golden``).
"""
from __future__ import annotations

import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from anyv2v_amd import ops  # noqa: E402

DEV = "cuda"
_GLDS_DEFAULT = ops.USE_GLDS


def _effective_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


# the CPU oracle: never more torch threads than the cgroup CPU quota (the GPU box: 256 hw threads, quota 16)
torch.set_num_threads(min(torch.get_num_threads(), _effective_cpus()))


def _rel(a: torch.Tensor, b: torch.Tensor):
    a = a.float()
    b = b.float().to(a.device)
    denom = b.abs().max().clamp_min(1e-6)
    return float((a - b).abs().max() / denom), float((a - b).norm() / b.norm().clamp_min(1e-12))


def _res(name, got, ref, tol):
    if not torch.isfinite(got.float()).all():
        return dict(name=name, err=float("nan"), l2=float("nan"), tol=tol, ok=False)
    mx, l2 = _rel(got, ref)
    # both metrics are gated: max |diff| / max |ref| and relative L2 (which is never larger than sqrt(N) x the former but is the one
    # a systematic small error moves)
    return dict(name=name, err=mx, l2=l2, tol=tol, ok=bool(mx <= tol and l2 <= tol))


# Kernel-level tolerance: max |got - ref| / max |ref| against PyTorch fp32 on identical fp16 inputs.  The measured worst case over
# all kernel checks is 8.5e-4 (fp16 output rounding, 2^-11 of the largest magnitude, plus fp32 accumulation-order effects;
# tools/gpu_check.py, round 2) -- the bound is ~2.3 x that, so a 3 x regression fails.  (Round 1 used 4e-3 / 6e-3.)
KTOL = 2e-3


def rnd(*shape, scale=1.0, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) % 100000))
    return (torch.randn(*shape, generator=g) * scale).to(torch.float16).to(DEV)


# ------------------------------------------------------------------------------------------------ layouts
def check_selftest():
    out = []
    scratch = torch.zeros(16384, dtype=torch.uint8, device=DEV)
    g = torch.Generator().manual_seed(5)
    A16 = torch.randint(-4, 5, (16, 32), generator=g).to(torch.float16)
    B16 = torch.randint(-4, 5, (32, 16), generator=g).to(torch.float16)
    A32 = torch.randint(-4, 5, (32, 16), generator=g).to(torch.float16)
    B32 = torch.randint(-4, 5, (16, 32), generator=g).to(torch.float16)
    host = scratch.cpu()
    host[0:1024] = A16.view(torch.uint8).reshape(-1)
    host[1024:2048] = B16.view(torch.uint8).reshape(-1)
    host[3072:4096] = A32.view(torch.uint8).reshape(-1)
    host[4096:5120] = B32.view(torch.uint8).reshape(-1)
    scratch.copy_(host)
    ops.selftest(scratch)
    torch.cuda.synchronize()
    h = scratch.cpu()
    D16 = h[2048:3072].view(torch.float32).reshape(16, 16)
    D32 = h[5120:9216].view(torch.float32).reshape(32, 32)
    TR = h[9216:9728].view(torch.float16).reshape(64, 4)
    out.append(_res("mfma_16x16x32_f16 layout", D16, A16.float() @ B16.float(), 0.0))
    out.append(_res("mfma_32x32x16_f16 layout", D32, A32.float() @ B32.float(), 0.0))
    # expected tr16 semantics: within each 16-lane group, out[c][j] = in[4 j + (c >> 2)][c & 3]; in[i][e] = 4 (16 g + i) + e
    exp = torch.empty(64, 4)
    for l in range(64):
        gq, c = l // 16, l % 16
        for j in range(4):
            i = 4 * j + (c >> 2)
            exp[l, j] = 4 * (16 * gq + i) + (c & 3)
    r = _res("ds_read_b64_tr_b16 semantics (probe)", TR.float(), exp, 0.0)
    r["dump"] = TR.float()[:20].tolist()
    r["informational"] = True
    out.append(r)
    return out


# ------------------------------------------------------------------------------------------------ gemm
def _gemm_ref(a, w, bias=None, rowvec=None, rowvec_div=0, residual=None, act=0):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    if rowvec is not None:
        idx = torch.arange(y.shape[0], device=y.device) // rowvec_div
        y = y + rowvec.float()[idx]
    if act == ops.ACT_SILU:
        y = F.silu(y)
    elif act == ops.ACT_GELU:
        y = F.gelu(y)
    if residual is not None:
        y = y + residual.float()
    return y


def check_gemm(variants=("reg", "glds", "naive")):
    out = []
    cases = [  # (M, N, K)
        (300, 320, 320), (1000, 512, 512), (257, 64, 128), (128, 4, 320), (4096, 1280, 1280), (77, 640, 1024),
        # large M with ragged M / N tails, 1 and 2 K-tiles (128-row kernel; N = 320 / 640 may take the persistent one)
        (8300, 320, 320), (20001, 640, 1280), (9000, 512, 64), (8192, 4, 128), (12345, 160, 192),
    ]
    for var in variants:
        ops.USE_GLDS = var == "glds"
        naive = var == "naive"
        for (M, N, K) in cases:
            a, w = rnd(M, K), rnd(N, K, scale=1 / math.sqrt(K))
            bias, res = rnd(N), rnd(M, N)
            rv = rnd((M + 49) // 50, N)
            y = ops.gemm(a, w, bias=bias, rowvec=rv, rowvec_div=50, residual=res, naive=naive,
                         out=torch.zeros(M, (N + 7) // 8 * 8, dtype=torch.float16, device=DEV))[:, :N]
            out.append(_res(f"gemm[{var}] M{M} N{N} K{K} +bias+rowvec+res", y, _gemm_ref(a, w, bias, rv, 50, res), KTOL))
        # two sources + SiLU
        a0, a1 = rnd(500, 128), rnd(500, 64)
        w = rnd(320, 192, scale=0.1)
        y = ops.gemm(a0, w, a1=a1, act=ops.ACT_SILU, naive=naive)
        out.append(_res(f"gemm[{var}] two-source + silu", y, _gemm_ref(torch.cat([a0, a1], 1), w, act=ops.ACT_SILU), KTOL))
        # strided A / C views (column windows of wider buffers)
        big = rnd(400, 960)
        w = rnd(320, 320, scale=0.06)
        dst = torch.zeros(400, 640, dtype=torch.float16, device=DEV)
        ops.gemm(big[:, 320:640], w, out=dst[:, 320:], naive=naive)
        out.append(_res(f"gemm[{var}] strided views", dst[:, 320:], _gemm_ref(big[:, 320:640], w), KTOL))
        # GEGLU at small and large M
        for M in (333, 9001):
            dim, inner = 128, 512
            a = rnd(M, dim)
            wfull, bfull = rnd(2 * inner, dim, scale=1 / math.sqrt(dim)), rnd(2 * inner, scale=0.1)
            wh, wg = wfull[:inner].view(inner // 16, 16, dim), wfull[inner:].view(inner // 16, 16, dim)
            wp = torch.stack([wh, wg], 1).reshape(2 * inner, dim).contiguous()
            bp = torch.stack([bfull[:inner].view(-1, 16), bfull[inner:].view(-1, 16)], 1).reshape(-1).contiguous()
            y = ops.gemm(a, wp, bias=bp, act=ops.ACT_GEGLU, naive=naive)
            proj = a.float() @ wfull.float().t() + bfull.float()
            ref = proj[:, :inner] * F.gelu(proj[:, inner:])
            out.append(_res(f"gemm[{var}] GEGLU M{M}", y, ref, KTOL))
    ops.USE_GLDS = _GLDS_DEFAULT
    return out


def check_gemm_big(extra=0, tag="big", exact=True):
    """Persistent 256x320 kernel (gemm_big_kernel), forced with flag bit3 on shapes small enough for the references:
    single / multiple rounds per block, every mode, two-source K loop, bias / temb / residual / GEGLU epilogues.
    ``exact`` = False: the forced kernel sums K in a different order (stream-K: fp32 partial slabs) -- the bit-equality rows become
    5e-4 rows.  ``extra``: further AnyV2VGemmDesc.flags bits, e.g. bit17 (| bit19 / bit20) = the ping-pong kernel gemm_pp_kernel (192- / 256-row
    tiles) on every non-GEGLU case -- same references, same bit-equality with the 128-row kernel."""
    out = []
    saved = ops.GEMM_FLAGS
    ops.GEMM_FLAGS = (saved & ~4) | 8 | extra
    try:
        for (M, N, K, res, rvd) in [(512, 320, 320, True, 0), (1024, 640, 192, False, 128), (256, 960, 64, True, 256),
                                    (256 * 41, 2560, 128, True, 0), (256 * 300, 320, 64, False, 0)]:
            a, w, bias = rnd(M, K), rnd(N, K, scale=1 / math.sqrt(K)), rnd(N)
            r = rnd(M, N) if res else None
            rv = rnd(M // rvd, N) if rvd else None
            y = ops.gemm(a, w, bias=bias, rowvec=rv, rowvec_div=rvd, residual=r)
            out.append(_res(f"gemm[{tag}] M{M} N{N} K{K} res={res} rowvec={bool(rvd)}", y, _gemm_ref(a, w, bias, rv, rvd, r), KTOL))
            yn = ops.gemm(a, w, bias=bias, rowvec=rv, rowvec_div=rvd, residual=r, naive=True)
            out.append(_res(f"gemm[{tag}] == naive kernel M{M} N{N} K{K}", y, yn.float(), 2e-3))
            # the two tile-kernel families accumulate every output element in the same order: BIT-equal without split-K (what lets a
            # batch-hinted launch keep its own kernel family and only take the reference launch's split factor, gemm.hip dispatch)
            ops.GEMM_FLAGS = (saved & ~8) | 4 | 16
            ys = ops.gemm(a, w, bias=bias, rowvec=rv, rowvec_div=rvd, residual=r)
            ops.GEMM_FLAGS = (saved & ~4) | 8 | extra
            out.append(_res(f"gemm[{tag}] bit-equal to the 128-row kernel M{M} N{N} K{K}", y, ys.float(), 0.0 if exact else 5e-4))
        # two-source K loop (skip concat)
        a0, a1 = rnd(768, 128), rnd(768, 64)
        w = rnd(320, 192, scale=0.1)
        y = ops.gemm(a0, w, a1=a1)
        out.append(_res(f"gemm[{tag}] two-source", y, _gemm_ref(torch.cat([a0, a1], 1), w), KTOL))
        # GEGLU, one and several rounds
        for M in (512, 256 * 70):
            dim, inner = 128, 640
            a = rnd(M, dim)
            wfull, bfull = rnd(2 * inner, dim, scale=1 / math.sqrt(dim)), rnd(2 * inner, scale=0.1)
            wh, wg = wfull[:inner].view(inner // 16, 16, dim), wfull[inner:].view(inner // 16, 16, dim)
            wp = torch.stack([wh, wg], 1).reshape(2 * inner, dim).contiguous()
            bp = torch.stack([bfull[:inner].view(-1, 16), bfull[inner:].view(-1, 16)], 1).reshape(-1).contiguous()
            y = ops.gemm(a, wp, bias=bp, act=ops.ACT_GEGLU)
            proj = a.float() @ wfull.float().t() + bfull.float()
            out.append(_res(f"gemm[{tag}] GEGLU M{M}", y, proj[:, :inner] * F.gelu(proj[:, inner:]), KTOL))
        # rastered tile order of wide-N launches (8 x 4 super-tiles per XCD round; flags bits 13-16): every order computes each
        # output tile with the same K loop -> BIT-equal to the classic order; ragged M (holes in the last super-tile row),
        # plain and GEGLU epilogues, a forced order on a launch the auto rule would leave alone
        for (M, N, K, geglu) in [(192 * 40 + 70, 2560, 128, False), (192 * 16, 5120, 64, True), (192 * 33, 2560, 64, True)]:
            a, w, bias = rnd(M, K), rnd(N, K, scale=1 / math.sqrt(K)), rnd(N)
            act = ops.ACT_GEGLU if geglu else ops.ACT_NONE
            ops.GEMM_FLAGS = ((saved & ~4) | 8 | extra) | (1 << 13)
            y0 = ops.gemm(a, w, bias=bias, act=act)
            yn = ops.gemm(a, w, bias=bias, act=act, naive=True)
            out.append(_res(f"gemm[{tag}] classic order == naive kernel M{M} N{N} K{K} geglu={geglu}", y0, yn.float(), 2e-3))
            for code, nfast in ((0, 0), (2, 0), (3, 0), (3, 1), (4, 0), (5, 1), (6, 0)):
                ops.GEMM_FLAGS = ((saved & ~4) | 8 | extra) | (code << 13) | (nfast << 16)
                y = ops.gemm(a, w, bias=bias, act=act)
                out.append(_res(f"gemm[{tag}] raster code {code} nfast {nfast} bit-equal to the classic order M{M} N{N} geglu={geglu}", y,
                                y0.float(), 0.0 if exact else 5e-4))
            ops.GEMM_FLAGS = (saved & ~4) | 8 | extra
        # conv 3x3 (stride 1, stride 2, folded upsample) with temb row vector / residual
        n, ci, co, H, W = 8, 64, 320, 16, 16
        x, w, b = rnd(n, ci, H, W), rnd(co, ci, 3, 3, scale=1 / math.sqrt(9 * ci)), rnd(co)
        temb, res = rnd(4, co), rnd(n * H * W, co)
        y = ops.gemm(_to_tokens(x), _pack_conv(w), bias=b, rowvec=temb, rowvec_div=2 * H * W, residual=res,
                     mode=ops.MODE_CONV2D, conv=(H, W, H, W, 1, 0))
        ref = F.conv2d(x.float(), w.float(), b.float(), padding=1) + temb.float().repeat_interleave(2, 0)[:, :, None, None]
        out.append(_res(f"conv3x3[{tag}] s1 +bias+temb+res", y, _to_tokens(ref) + res.float(), KTOL))
        ops.GEMM_FLAGS = (saved & ~8) | 4 | 16
        ys = ops.gemm(_to_tokens(x), _pack_conv(w), bias=b, rowvec=temb, rowvec_div=2 * H * W, residual=res, mode=ops.MODE_CONV2D,
                      conv=(H, W, H, W, 1, 0))
        ops.GEMM_FLAGS = (saved & ~4) | 8 | extra
        out.append(_res(f"conv3x3[{tag}] bit-equal to the 128-row kernel", y, ys.float(), 0.0 if exact else 5e-4))
        y = ops.gemm(_to_tokens(x), _pack_conv(w), bias=b, mode=ops.MODE_CONV2D, conv=(H, W, H // 2, W // 2, 2, 0),
                     M=n * (H // 2) * (W // 2))
        out.append(_res(f"conv3x3[{tag}] stride 2", y, _to_tokens(F.conv2d(x.float(), w.float(), b.float(), stride=2, padding=1)), KTOL))
        y = ops.gemm(_to_tokens(x), _pack_conv(w), bias=b, mode=ops.MODE_CONV2D, conv=(H, W, 2 * H, 2 * W, 1, 1),
                     M=n * 4 * H * W)
        ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), b.float(), padding=1)
        out.append(_res(f"conv3x3[{tag}] nearest-x2 folded", y, _to_tokens(ref), KTOL))
        # two-source conv (up-block skip concat)
        x1 = rnd(n, 128, H, W)
        w2 = rnd(co, ci + 128, 3, 3, scale=1 / math.sqrt(9 * (ci + 128)))
        y = ops.gemm(_to_tokens(x), _pack_conv(w2), a1=_to_tokens(x1), bias=b, mode=ops.MODE_CONV2D, conv=(H, W, H, W, 1, 0))
        ref = F.conv2d(torch.cat([x, x1], 1).float(), w2.float(), b.float(), padding=1)
        out.append(_res(f"conv3x3[{tag}] two-source", y, _to_tokens(ref), KTOL))
        # temporal (3,1,1) conv with residual
        B_, Fr, HW, C = 2, 8, 64, 128
        xt = rnd(B_ * Fr * HW, C)
        wt, bt, rt = rnd(320, C, 3, scale=1 / math.sqrt(3 * C)), rnd(320), rnd(B_ * Fr * HW, 320)
        y = ops.gemm(xt, wt.permute(0, 2, 1).reshape(320, 3 * C).contiguous(), bias=bt, residual=rt, mode=ops.MODE_TEMPORAL,
                     temporal=(Fr, HW))
        x5 = xt.float().view(B_, Fr, HW, C).permute(0, 3, 1, 2)  # [B, C, F, HW]
        ref = F.conv1d(x5.permute(0, 3, 1, 2).reshape(B_ * HW, C, Fr), wt.float(), bt.float(), padding=1)
        ref = ref.view(B_, HW, 320, Fr).permute(0, 3, 1, 2).reshape(B_ * Fr * HW, 320) + rt.float()
        out.append(_res(f"temporal conv[{tag}] +res", y, ref, KTOL))
        # more output tiles than CUs AND several K-tiles per tile: the persistent loop's tile switch with a live LDS-DMA pipeline
        # (conv 3x3 + temb + residual over 72 images of 32 x 32: 288 / 384 tiles, 9 K-tiles; temporal conv: 300 / 400 tiles, 6 K-tiles)
        n, ci, co, H, W = 72, 64, 320, 32, 32
        x, w, b = rnd(n, ci, H, W), rnd(co, ci, 3, 3, scale=1 / math.sqrt(9 * ci)), rnd(co)
        temb, res = rnd(n // 8, co), rnd(n * H * W, co)
        for r_ in (res, None):
            y = ops.gemm(_to_tokens(x), _pack_conv(w), bias=b, rowvec=temb, rowvec_div=8 * H * W, residual=r_, mode=ops.MODE_CONV2D,
                         conv=(H, W, H, W, 1, 0))
            ref = F.conv2d(x.float(), w.float(), b.float(), padding=1) + temb.float().repeat_interleave(8, 0)[:, :, None, None]
            out.append(_res(f"conv3x3[{tag}] 288+ tiles x 9 K-tiles res={r_ is not None}", y, _to_tokens(ref) + (r_.float() if r_ is not None else 0.0), KTOL))
            ops.GEMM_FLAGS = (saved & ~8) | 4 | 16
            ys = ops.gemm(_to_tokens(x), _pack_conv(w), bias=b, rowvec=temb, rowvec_div=8 * H * W, residual=r_, mode=ops.MODE_CONV2D,
                          conv=(H, W, H, W, 1, 0))
            ops.GEMM_FLAGS = (saved & ~4) | 8 | extra
            out.append(_res(f"conv3x3[{tag}] 288+ tiles bit-equal to the 128-row kernel res={r_ is not None}", y, ys.float(), 0.0 if exact else 5e-4))
        B_, Fr, HW, C = 3, 16, 1600, 128
        xt = rnd(B_ * Fr * HW, C)
        wt, bt, rt = rnd(320, C, 3, scale=1 / math.sqrt(3 * C)), rnd(320), rnd(B_ * Fr * HW, 320)
        y = ops.gemm(xt, wt.permute(0, 2, 1).reshape(320, 3 * C).contiguous(), bias=bt, residual=rt, mode=ops.MODE_TEMPORAL, temporal=(Fr, HW))
        ref = F.conv1d(xt.float().view(B_, Fr, HW, C).permute(0, 2, 3, 1).reshape(B_ * HW, C, Fr), wt.float(), bt.float(), padding=1)
        ref = ref.view(B_, HW, 320, Fr).permute(0, 3, 1, 2).reshape(B_ * Fr * HW, 320) + rt.float()
        out.append(_res(f"temporal conv[{tag}] 300+ tiles x 6 K-tiles +res", y, ref, KTOL))
        a, w2, bb = rnd(256 * 300 + 40, 1280), rnd(640, 1280, scale=1 / math.sqrt(1280)), rnd(640)
        r2 = rnd(256 * 300 + 40, 640)
        y = ops.gemm(a, w2, bias=bb, residual=r2)
        out.append(_res(f"gemm[{tag}] FF-down shape 76840 x 640 x 1280 +res (ragged M)", y, _gemm_ref(a, w2, bb, None, 0, r2), KTOL))
        ops.GEMM_FLAGS = (saved & ~8) | 4 | 16
        ys = ops.gemm(a, w2, bias=bb, residual=r2)
        ops.GEMM_FLAGS = (saved & ~4) | 8 | extra
        out.append(_res(f"gemm[{tag}] FF-down shape bit-equal to the 128-row kernel", y, ys.float(), 0.0 if exact else 5e-4))
    finally:
        ops.GEMM_FLAGS = saved
    return out


def check_conv_halo():
    """gemm_swh_kernel (round 6: 3x3 stride-1 convolution whose A operand is staged ONCE per (dy, channel slice) as an LDS patch and
    read three times with a one-pixel address shift, K order (dy, slice, dx)), forced with flags bit28 on every image width it
    takes (16 / 32 / 64): one / two sources, bias / temb row vector / residual epilogues, tile counts below and above the CU count,
    row counts that are not a multiple of the 192-row tile (images cut by a tile boundary), against F.conv2d in fp32 (kernel tolerance)
    and against the tap-gather kernels' result (5e-4: the same products in another summation order)."""
    out = []
    saved = ops.GEMM_FLAGS
    try:
        for (n, H, c0, c1, co, res, rvd) in [(3, 64, 64, 0, 320, False, 0), (2, 64, 128, 64, 320, True, 0), (6, 32, 128, 0, 640, False, 2),
                                             (70, 32, 64, 0, 320, True, 0), (5, 16, 192, 0, 320, False, 0), (13, 16, 64, 64, 640, True, 0),
                                             (300, 16, 64, 0, 320, False, 100)]:
            x0 = rnd(n, c0, H, H)
            x1 = rnd(n, c1, H, H) if c1 else None
            ci = c0 + c1
            w, b = rnd(co, ci, 3, 3, scale=1 / math.sqrt(9 * ci)), rnd(co)
            r = rnd(n * H * H, co) if res else None
            temb = rnd(n // rvd, co) if rvd else None
            kw = dict(bias=b, a1=None if x1 is None else _to_tokens(x1), rowvec=temb, rowvec_div=rvd * H * H if rvd else 0, residual=r,
                      mode=ops.MODE_CONV2D, conv=(H, H, H, H, 1, 0))
            ops.GEMM_FLAGS = saved | (1 << 28)
            y = ops.gemm(_to_tokens(x0), _pack_conv(w), **kw)
            ops.GEMM_FLAGS = saved
            yd = ops.gemm(_to_tokens(x0), _pack_conv(w), **kw)
            xx = x0 if x1 is None else torch.cat([x0, x1], 1)
            ref = F.conv2d(xx.float(), w.float(), b.float(), padding=1)
            if rvd:
                ref = ref + temb.float().repeat_interleave(rvd, 0)[:, :, None, None]
            ref = _to_tokens(ref) + (r.float() if res else 0.0)
            tag = f"{n} x {H}x{H}, {c0}+{c1} -> {co}, res={res} temb={bool(rvd)}"
            out.append(_res(f"conv3x3[halo] {tag} vs F.conv2d fp32", y, ref, KTOL))
            out.append(_res(f"conv3x3[halo] {tag} vs the tap-gather kernels", y, yd.float(), 5e-4))
    finally:
        ops.GEMM_FLAGS = saved
    return out


def _geglu_pack(wfull, bfull, inner):
    dim = wfull.shape[1]
    wh, wg = wfull[:inner].view(inner // 16, 16, dim), wfull[inner:].view(inner // 16, 16, dim)
    wp = torch.stack([wh, wg], 1).reshape(2 * inner, dim).contiguous()
    bp = torch.stack([bfull[:inner].view(-1, 16), bfull[inner:].view(-1, 16)], 1).reshape(-1).contiguous()
    return wp, bp


def check_gemm_ws():
    """Weight-stationary K = 320 kernel (gemm_ws.hip), forced with flag bit10 at sizes the references finish quickly, and
    through the dispatch threshold at the bench's own row counts: one / several strips per wave, ragged M (row guards and
    clamped look-ahead), 1 / 2 / 6 / 16 slabs, bias / residual / GEGLU epilogues, strided A / C / R views, and equality
    with the tile kernels it replaces (flag bit9)."""
    out = []
    saved = ops.GEMM_FLAGS
    K = 320
    try:
        for (M, N, res) in [(100, 160, False), (2049, 320, True), (4113, 960, False), (9000, 320, True), (33000, 320, True),
                            (70001, 960, False)]:
            a, w, bias = rnd(M, K), rnd(N, K, scale=1 / math.sqrt(K)), rnd(N)
            r = rnd(M, N) if res else None
            ops.GEMM_FLAGS = saved | 1024
            y = ops.gemm(a, w, bias=bias, residual=r)
            out.append(_res(f"gemm[ws] M{M} N{N} res={res}", y, _gemm_ref(a, w, bias, residual=r), KTOL))
            ops.GEMM_FLAGS = saved | 512
            yt = ops.gemm(a, w, bias=bias, residual=r)
            out.append(_res(f"gemm[ws] == tile kernels M{M} N{N}", y, yt.float(), 1e-3))
        # no bias; column windows of wider buffers for A, C and R (the V-only projection writes qkv[:, 640:960])
        ops.GEMM_FLAGS = saved | 1024
        big, w = rnd(5000, 960), rnd(320, K, scale=1 / math.sqrt(K))
        dst = torch.zeros(5000, 960, dtype=torch.float16, device=DEV)
        rbuf = rnd(5000, 640)
        ops.gemm(big[:, 320:640], w, out=dst[:, 640:], residual=rbuf[:, 320:])
        out.append(_res("gemm[ws] strided A / C / R views", dst[:, 640:], _gemm_ref(big[:, 320:640], w, residual=rbuf[:, 320:]), KTOL))
        out.append(_res("gemm[ws] strided C leaves the other columns alone", dst[:, :640], torch.zeros(5000, 640), 0.0))
        # GEGLU: 2 slabs (inner 160) and 16 slabs (inner 1280, the layer's own width)
        for (M, inner) in [(777, 160), (3000, 1280), (40000, 1280)]:
            a = rnd(M, K)
            wfull, bfull = rnd(2 * inner, K, scale=1 / math.sqrt(K)), rnd(2 * inner, scale=0.1)
            wp, bp = _geglu_pack(wfull, bfull, inner)
            y = ops.gemm(a, wp, bias=bp, act=ops.ACT_GEGLU)
            proj = a.float() @ wfull.float().t() + bfull.float()
            out.append(_res(f"gemm[ws] GEGLU M{M} inner{inner}", y, proj[:, :inner] * F.gelu(proj[:, inner:]), KTOL))
        # transformer_in's width (K = 512): 64-column slabs (128 for GEGLU), 16 K-steps, ring of 8
        ops.GEMM_FLAGS = saved | 1024
        K5 = 512
        for (M, N, res) in [(4100, 512, True), (9001, 512, False), (33000, 512, True)]:
            a, w, bias = rnd(M, K5), rnd(N, K5, scale=1 / math.sqrt(K5)), rnd(N)
            r = rnd(M, N) if res else None
            ops.GEMM_FLAGS = saved | 1024
            y = ops.gemm(a, w, bias=bias, residual=r)
            out.append(_res(f"gemm[ws] K512 M{M} N{N} res={res}", y, _gemm_ref(a, w, bias, residual=r), KTOL))
            ops.GEMM_FLAGS = saved | 512
            out.append(_res(f"gemm[ws] K512 == tile kernels M{M} N{N}", y, ops.gemm(a, w, bias=bias, residual=r).float(), 1e-3))
        ops.GEMM_FLAGS = saved | 1024
        for (M, inner) in [(3000, 2048), (777, 128)]:
            a = rnd(M, K5)
            wfull, bfull = rnd(2 * inner, K5, scale=1 / math.sqrt(K5)), rnd(2 * inner, scale=0.1)
            wp, bp = _geglu_pack(wfull, bfull, inner)
            y = ops.gemm(a, wp, bias=bp, act=ops.ACT_GEGLU)
            proj = a.float() @ wfull.float().t() + bfull.float()
            out.append(_res(f"gemm[ws] K512 GEGLU M{M} inner{inner}", y, proj[:, :inner] * F.gelu(proj[:, inner:]), KTOL))
        # the dispatch threshold itself (no flag): the inversion step's row count, with residual; bit-reproducible
        ops.GEMM_FLAGS = saved
        M = 65536
        a, w, bias, r = rnd(M, K), rnd(320, K, scale=1 / math.sqrt(K)), rnd(320), rnd(M, 320)
        y = ops.gemm(a, w, bias=bias, residual=r)
        out.append(_res("gemm[ws] dispatch M65536 N320 +res", y, _gemm_ref(a, w, bias, residual=r), KTOL))
        out.append(_res("gemm[ws] bit-reproducible", y, ops.gemm(a, w, bias=bias, residual=r).float(), 0.0))
        ops.GEMM_FLAGS = saved | 512
        out.append(_res("gemm[ws] dispatch == tile kernels", y, ops.gemm(a, w, bias=bias, residual=r).float(), 1e-3))
    finally:
        ops.GEMM_FLAGS = saved
    return out


def check_gemm_ws_ln():
    """LayerNorm folded into the weight-stationary kernel (``ops.gemm(..., ln=...)``): un-normalised rows in, gamma / beta folded
    into the weights (``ops.ln_fold``), row statistics taken by the matrix pipe.  Against torch fp32 LayerNorm -> Linear (-> GEGLU)
    on the same fp16 inputs, and against the unfused HIP path (layernorm kernel + GEMM); rows with a large mean (|mean| = 6 sigma:
    the fold subtracts mean x column sums) and ragged row counts included."""
    out = []
    eps = 1e-5
    for (M, K, N, act, shift) in [(40000, 320, 960, 0, 0.0), (33001, 320, 320, 0, 6.0), (36000, 320, 2560, ops.ACT_GEGLU, 0.0),
                                  (34000, 512, 4096, ops.ACT_GEGLU, 3.0)]:
        x = (rnd(M, K).float() * (1.0 + 0.5 * torch.rand(M, 1, device=DEV)) + shift * torch.randn(M, 1, device=DEV)).half()
        gamma, beta = (1.0 + 0.3 * rnd(K).float()).half(), (0.2 * rnd(K, seed=7).float()).half()
        geglu = act == ops.ACT_GEGLU
        wfull = rnd(N, K, scale=1 / math.sqrt(K))
        bfull = rnd(N, scale=0.1) if geglu or N != 960 else None      # attention's to_q / to_k / to_v have no bias
        y_ln = F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), eps)
        proj = y_ln @ wfull.float().t() + (bfull.float() if bfull is not None else 0.0)
        if geglu:
            inner = N // 2
            ref = proj[:, :inner] * F.gelu(proj[:, inner:])
            wp, bp = _geglu_pack(wfull, bfull, inner)
        else:
            ref, wp, bp = proj, wfull, bfull
        wq, bq, c1 = ops.ln_fold(wp, bp, gamma, beta)
        assert ops.ln_gemm_supported(M, K, N, act)
        y = ops.gemm(x, wq, bias=bq, act=act, ln=(c1, eps))
        out.append(_res(f"gemm[ws+LN] M{M} K{K} N{N} act{act} mean-shift {shift}: vs torch fp32 LayerNorm -> Linear", y, ref, KTOL))
        h = ops.layernorm(x, gamma, beta, eps)
        yu = ops.gemm(h, wp, bias=bp, act=act)
        out.append(_res(f"gemm[ws+LN] M{M} K{K} N{N}: vs layernorm kernel + GEMM", y, yu.float(), 3e-3))
    # shapes outside the fold's coverage are refused loudly, not silently computed without the LayerNorm
    try:
        ops.gemm(rnd(64, 640), rnd(640, 640), ln=(torch.zeros(640, device=DEV), eps))
        out.append(dict(name="gemm[ws+LN] unsupported shape is refused", err=1.0, l2=1.0, tol=0.0, ok=False))
    except Exception as e:
        out.append(dict(name=f"gemm[ws+LN] unsupported shape is refused ({type(e).__name__})", err=0.0, l2=0.0, tol=0.0, ok=True))
    return out


def check_ff_fused():
    """anyv2v_ff_geglu_f16 (ff_fused.hip): GEGLU up-projection + down-projection (+ residual) of the 320-channel blocks in one kernel vs
    torch fp32 on the un-packed weights (hidden rounded to fp16 once, as the unfused GEGLU GEMM stores it), and vs the unfused pair of
    GEMMs; one strip, ragged tails, more strips than the 1024 pair slots (several rounds: the weight rings wrap), with / without residual."""
    out = []
    C, H = 320, 1280
    w1, b1 = rnd(2 * H, C, scale=1 / math.sqrt(C)), rnd(2 * H, scale=0.1)
    w2, b2 = rnd(C, H, scale=1 / math.sqrt(H)), rnd(C, scale=0.1)
    w1p, b1p = _geglu_pack(w1, b1, H)
    w2s = ops.ff_pack_w2(w2)
    for M, res in [(32, False), (32 * 5 + 7, True), (128 * 256, True), (128 * 256 * 2 + 32 * 3 + 5, True), (196608, False)]:
        x = rnd(M, C)
        r = rnd(M, C) if res else None
        y = ops.ff_geglu(x, w1p, b1p, w2s, b2, residual=r)
        proj = x.float() @ w1.float().t() + b1.float()
        hid = (proj[:, :H] * F.gelu(proj[:, H:])).half().float()
        ref = (hid @ w2.float().t() + b2.float()).half().float()
        if res:
            ref = ref + r.float()
        out.append(_res(f"ff_fused M{M} res={res} vs torch fp32", y, ref, KTOL))
        g = ops.gemm(x, w1p, bias=b1p, act=ops.ACT_GEGLU)
        yu = ops.gemm(g, w2, bias=b2, residual=r)
        out.append(_res(f"ff_fused M{M} res={res} vs the unfused GEGLU GEMM + Linear", y, yu.float(), KTOL))
    return out


def check_gemm_splitk():
    """Split-K path (launches that cannot fill the chip and have >= 16 K-tiles; fp32 partial tiles + a second pass that
    sums them in split order): against torch, against the unsplit kernel (flag bit4), and bit-reproducibility.
    (A fused last-arriver reduction was tried and dropped: the device-scope fences it needs write back / invalidate the
    per-XCD L2 on gfx950 and cost far more than the second launch -- DESIGN.md.)"""
    out = []
    saved = ops.GEMM_FLAGS
    try:
        # 8x8-level conv: M = 3 clips x 16 frames x 64 px, K = 2304 (36 K-tiles), 24 x 2 = 48 tiles -> 4 splits
        n, ci, co, H, W = 48, 256, 320, 8, 8
        x, w, b = rnd(n, ci, H, W), rnd(co, ci, 3, 3, scale=1 / math.sqrt(9 * ci)), rnd(co)
        temb, res = rnd(3, co), rnd(n * H * W, co)
        kw = dict(bias=b, rowvec=temb, rowvec_div=16 * H * W, residual=res, mode=ops.MODE_CONV2D, conv=(H, W, H, W, 1, 0))
        ref = F.conv2d(x.float(), w.float(), b.float(), padding=1) + temb.float().repeat_interleave(16, 0)[:, :, None, None]
        ref = _to_tokens(ref) + res.float()
        xt, wp = _to_tokens(x), _pack_conv(w)
        ys = [ops.gemm(xt, wp, **kw) for _ in range(3)]
        out.append(_res("conv3x3[split-K] vs torch", ys[0], ref, KTOL))
        out.append(_res("conv3x3[split-K] bit-reproducible", ys[0], ys[2].float(), 0.0))
        ops.GEMM_FLAGS = saved | 16
        y_one = ops.gemm(xt, wp, **kw)
        out.append(_res("conv3x3[split-K] vs unsplit kernel", ys[0], y_one.float(), 2e-3))
        ops.GEMM_FLAGS = saved
        # same level, N = 1280: split-K work items of the persistent 192x320 kernel (16 x 4 tiles x 4 splits = one round);
        # M = 3000 is not a multiple of 192 (row guards), compared with the 128-row kernel's split path (flag bit2) too
        n, ci, co = 48, 640, 1280
        x, w, b = rnd(n, ci, H, W), rnd(co, ci, 3, 3, scale=1 / math.sqrt(9 * ci)), rnd(co)
        temb, res = rnd(3, co), rnd(n * H * W, co)
        kw = dict(bias=b, rowvec=temb, rowvec_div=16 * H * W, residual=res, mode=ops.MODE_CONV2D, conv=(H, W, H, W, 1, 0))
        ref = F.conv2d(x.float(), w.float(), b.float(), padding=1) + temb.float().repeat_interleave(16, 0)[:, :, None, None]
        ref = _to_tokens(ref) + res.float()
        xt, wp = _to_tokens(x), _pack_conv(w)
        ys = [ops.gemm(xt, wp, **kw) for _ in range(3)]
        out.append(_res("conv3x3[persistent split-K] vs torch", ys[0], ref, KTOL))
        out.append(_res("conv3x3[persistent split-K] bit-reproducible", ys[0], ys[2].float(), 0.0))
        ops.GEMM_FLAGS = saved | 4
        out.append(_res("conv3x3[persistent split-K] vs 128-row split-K", ys[0], ops.gemm(xt, wp, **kw).float(), 2e-3))
        ops.GEMM_FLAGS = saved
        a, wl, rl = rnd(3000, 5120), rnd(1280, 5120, scale=1 / math.sqrt(5120)), rnd(3000, 1280)
        y = ops.gemm(a, wl, bias=rnd(1280) * 0, residual=rl)
        out.append(_res("gemm[persistent split-K] M3000 N1280 K5120 +res", y, _gemm_ref(a, wl) + rl.float(), KTOL))
        # linear, long K, small M
        a, wl = rnd(1024, 5120), rnd(1280, 5120, scale=1 / math.sqrt(5120))
        y = ops.gemm(a, wl, bias=rnd(1280) * 0)
        out.append(_res("gemm[split-K] M1024 N1280 K5120", y, _gemm_ref(a, wl), KTOL))
    finally:
        ops.GEMM_FLAGS = saved
    return out


def _to_tokens(x):  # NCHW -> [(n h w), c]
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous()


def _pack_conv(w):  # [Co, Ci, 3, 3] -> [Co, 9*Ci]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def check_conv(variants=("reg", "glds", "naive")):
    out = []
    for var in variants:
        ops.USE_GLDS = var == "glds"
        naive = var == "naive"
        # 3x3 stride 1 with bias + temb rowvec + residual
        n, ci, co, H, W = 6, 64, 128, 12, 10
        x, w, b = rnd(n, ci, H, W), rnd(co, ci, 3, 3, scale=1 / math.sqrt(9 * ci)), rnd(co)
        temb, res = rnd(3, co), rnd(n * H * W, co)
        y = ops.gemm(_to_tokens(x), _pack_conv(w), bias=b, rowvec=temb, rowvec_div=2 * H * W, residual=res,
                     mode=ops.MODE_CONV2D, conv=(H, W, H, W, 1, 0), naive=naive)
        ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
        ref = ref + temb.float().repeat_interleave(2, 0)[:, :, None, None]
        ref = _to_tokens(ref) + res.float()
        out.append(_res(f"conv3x3[{var}] s1 +bias+temb+res", y, ref, KTOL))
        # stride 2
        Ho, Wo = H // 2, W // 2
        y = ops.gemm(_to_tokens(x), _pack_conv(w), bias=b, mode=ops.MODE_CONV2D, conv=(H, W, Ho, Wo, 2, 0),
                     M=n * Ho * Wo, naive=naive)
        ref = _to_tokens(F.conv2d(x.float(), w.float(), b.float(), stride=2, padding=1))
        out.append(_res(f"conv3x3[{var}] stride 2", y, ref, KTOL))
        # nearest x2 upsample folded
        y = ops.gemm(_to_tokens(x), _pack_conv(w), bias=b, mode=ops.MODE_CONV2D, conv=(H, W, 2 * H, 2 * W, 1, 1),
                     M=n * 4 * H * W, naive=naive)
        ref = _to_tokens(F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), b.float(), padding=1))
        out.append(_res(f"conv3x3[{var}] upsample x2", y, ref, KTOL))
        # two sources (skip concat)
        x1 = rnd(n, 128, H, W)
        w2 = rnd(co, ci + 128, 3, 3, scale=1 / math.sqrt(9 * (ci + 128)))
        y = ops.gemm(_to_tokens(x), _pack_conv(w2), a1=_to_tokens(x1), mode=ops.MODE_CONV2D, conv=(H, W, H, W, 1, 0), naive=naive)
        ref = _to_tokens(F.conv2d(torch.cat([x, x1], 1).float(), w2.float(), padding=1))
        out.append(_res(f"conv3x3[{var}] two-source", y, ref, KTOL))
        # temporal (3,1,1)
        B, Fr, HW, C = 2, 5, 24, 64
        xt = rnd(B * Fr * HW, C)
        w3, b3 = rnd(C, C, 3, 1, 1, scale=1 / math.sqrt(3 * C)), rnd(C)
        wp = w3[:, :, :, 0, 0].permute(0, 2, 1).reshape(C, -1).contiguous()
        y = ops.gemm(xt, wp, bias=b3, mode=ops.MODE_TEMPORAL, temporal=(Fr, HW), residual=xt, naive=naive)
        x5 = xt.view(B, Fr, HW, C).permute(0, 3, 1, 2).unsqueeze(-1).float()  # [B,C,F,HW,1]
        ref = F.conv3d(x5, w3.float(), b3.float(), padding=(1, 0, 0)) + x5
        ref = ref.squeeze(-1).permute(0, 2, 3, 1).reshape(B * Fr * HW, C)
        out.append(_res(f"temporal conv[{var}] +res", y, ref, KTOL))
        # large M (>= 8192 rows): conv s1 two-source + temb + res, upsample, temporal
        n, ci, c1, co, H, W = 6, 64, 128, 160, 40, 36
        x, x1 = rnd(n, ci, H, W), rnd(n, c1, H, W)
        w2, b = rnd(co, ci + c1, 3, 3, scale=1 / math.sqrt(9 * (ci + c1))), rnd(co)
        temb, res = rnd(3, co), rnd(n * H * W, co)
        y = ops.gemm(_to_tokens(x), _pack_conv(w2), a1=_to_tokens(x1), bias=b, rowvec=temb, rowvec_div=2 * H * W,
                     residual=res, mode=ops.MODE_CONV2D, conv=(H, W, H, W, 1, 0), naive=naive)
        ref = F.conv2d(torch.cat([x, x1], 1).float(), w2.float(), b.float(), padding=1)
        ref = _to_tokens(ref + temb.float().repeat_interleave(2, 0)[:, :, None, None]) + res.float()
        out.append(_res(f"conv3x3[{var}] large-M two-source +bias+temb+res", y, ref, KTOL))
        xs = rnd(3, 64, 30, 30)
        ws = rnd(320, 64, 3, 3, scale=1 / math.sqrt(9 * 64))
        y = ops.gemm(_to_tokens(xs), _pack_conv(ws), mode=ops.MODE_CONV2D, conv=(30, 30, 60, 60, 1, 1), M=3 * 3600, naive=naive)
        ref = _to_tokens(F.conv2d(F.interpolate(xs.float(), scale_factor=2.0, mode="nearest"), ws.float(), padding=1))
        out.append(_res(f"conv3x3[{var}] large-M upsample x2", y, ref, KTOL))
        B, Fr, HW, C = 2, 8, 600, 128
        xt = rnd(B * Fr * HW, C)
        w3, b3 = rnd(C, C, 3, 1, 1, scale=1 / math.sqrt(3 * C)), rnd(C)
        wp = w3[:, :, :, 0, 0].permute(0, 2, 1).reshape(C, -1).contiguous()
        y = ops.gemm(xt, wp, bias=b3, mode=ops.MODE_TEMPORAL, temporal=(Fr, HW), residual=xt, naive=naive)
        x5 = xt.view(B, Fr, HW, C).permute(0, 3, 1, 2).unsqueeze(-1).float()
        ref = (F.conv3d(x5, w3.float(), b3.float(), padding=(1, 0, 0)) + x5).squeeze(-1).permute(0, 2, 3, 1).reshape(B * Fr * HW, C)
        out.append(_res(f"temporal conv[{var}] large-M +res", y, ref, KTOL))
    # tiny-channel conv goes through the reference-grade kernel automatically
    x, w, b = rnd(2, 4, 8, 8), rnd(16, 4, 3, 3, scale=0.2), rnd(16)
    y = ops.gemm(_to_tokens(x), _pack_conv(w), bias=b, act=ops.ACT_SILU, mode=ops.MODE_CONV2D, conv=(8, 8, 8, 8, 1, 0))
    out.append(_res("conv3x3 Cin=4 (naive path) + silu", y, _to_tokens(F.silu(F.conv2d(x.float(), w.float(), b.float(), padding=1))), KTOL))
    ops.USE_GLDS = _GLDS_DEFAULT
    return out


# ------------------------------------------------------------------------------------------------ norms
def check_norms():
    out = []
    stats = torch.zeros(ops.gn_scratch_floats(64, 1), dtype=torch.float32, device=DEV)
    for (n, c, hw, silu, eps) in [(6, 320, 100, True, 1e-5), (3, 64, 64, False, 1e-6), (4, 1280, 64, True, 1e-5)]:
        x = rnd(n * hw, c) + 0.5
        ga, be = rnd(c) + 1.0, rnd(c)
        y = ops.groupnorm(x, ga, be, stats, hw, groups=32, eps=eps, silu=silu)
        ref = F.group_norm(x.float().view(n, hw, c).permute(0, 2, 1), 32, ga.float(), be.float(), eps)
        if silu:
            ref = F.silu(ref)
        out.append(_res(f"groupnorm 4-D n{n} c{c} hw{hw} silu={silu}", y, ref.permute(0, 2, 1).reshape(n * hw, c), KTOL))
    # 5-D (per clip over frames) + two sources with a group straddling the boundary (1280 + 640 -> 60 ch/group)
    B, Fr, hw, c0, c1 = 2, 3, 16, 1280, 640
    x0, x1 = rnd(B * Fr * hw, c0), rnd(B * Fr * hw, c1) * 2 + 1
    ga, be = rnd(c0 + c1) + 1.0, rnd(c0 + c1)
    y = ops.groupnorm(x0, ga, be, stats, Fr * hw, x1=x1, groups=32, eps=1e-5, silu=True)
    xc = torch.cat([x0, x1], 1).float().view(B, Fr * hw, c0 + c1).permute(0, 2, 1)
    ref = F.silu(F.group_norm(xc, 32, ga.float(), be.float(), 1e-5)).permute(0, 2, 1).reshape(B * Fr * hw, c0 + c1)
    out.append(_res("groupnorm 5-D two-source silu", y, ref, KTOL))
    # sharded statistics (frame-parallel clips): partial + apply(shards=1) IS the one-call kernel pair (bit-identical);
    # a clip split into two pixel halves, partial sums added as the all-reduce would, equals the unsharded result
    y1 = ops.groupnorm(x0, ga, be, stats, Fr * hw, x1=x1, groups=32, eps=1e-5, silu=True, shard=(1, lambda t: t))
    out.append(_res("groupnorm two-phase (shards=1) == one call", y1, y.float(), 0.0 if DEV != "cpu" else 2e-3))
    xs = x0.view(B, Fr, hw, c0)
    halves = [xs[:, :, i * hw // 2:(i + 1) * hw // 2].reshape(-1, c0).contiguous() for i in range(2)]
    ga0, be0 = ga[:c0].contiguous(), be[:c0].contiguous()
    full = ops.groupnorm(x0, ga0, be0, stats, Fr * hw, groups=32, eps=1e-5, silu=True).view(B, Fr, hw, c0)
    st = [torch.zeros_like(stats) for _ in range(2)]
    ops.groupnorm(halves[1], ga0, be0, st[1], Fr * hw // 2, groups=32, silu=True, shard=(1, lambda t: t))  # rank 1's sums
    other = st[1].clone()
    got = ops.groupnorm(halves[0], ga0, be0, st[0], Fr * hw // 2, groups=32, eps=1e-5, silu=True,
                        shard=(2, lambda t: t.add_(other[:t.numel()])))
    out.append(_res("groupnorm sharded over 2 pixel halves vs unsharded", got,
                    full[:, :, :hw // 2].reshape(-1, c0).float(), 2e-3))
    for (m, c) in [(1000, 320), (77, 1280), (333, 512), (50, 4), (64, 64)]:
        x = rnd(m, c) * 2 + 0.3
        ga, be = rnd(c) + 1.0, rnd(c)
        y = ops.layernorm(x, ga, be, 1e-5)
        out.append(_res(f"layernorm m{m} c{c}", y, F.layer_norm(x.float(), (c,), ga.float(), be.float(), 1e-5), KTOL))
    return out


# ------------------------------------------------------------------------------------------------ attention
def _sdpa(q, k, v):  # [b, h, s, d] fp32
    return F.scaled_dot_product_attention(q.float(), k.float(), v.float())


def check_attention_forced_rescale():
    """The flash kernel exponentiates every 16-key step against the OLD running maximum and uses the partial row sum as the
    range check; the textbook path (true maximum, rescale of O and l, re-exponentiation) only runs when that check trips and
    on the very first step.  Bounded random data never takes the branch after the first step, so it is forced here (MI355X
    guide rule 26): spiked keys at chosen positions (start / middle of a tile, last key, ragged tail), a first step whose
    scores sit far BELOW the rest (maximum has to climb), far ABOVE the rest (everything after underflows), and scores that
    overflow fp32 exp2 against the old maximum (inf in the check).  Reference: fp32 SDPA on the same fp16 inputs."""
    out = []
    for name, S, Sk, build in (
        ("spike at key 70 (2nd tile, 1st step)", 256, 256, lambda q, k: _spike(q, k, [70], 6.0)),
        ("spikes at keys 5, 200, 255 (every path of a tile)", 256, 256, lambda q, k: _spike(q, k, [5, 200, 255], 5.0)),
        ("spike in the ragged tail (key 144 of 145)", 128, 145, lambda q, k: _spike(q, k, [144], 6.0)),
        ("first 16 keys far below the rest", 128, 512, lambda q, k: _shift_first(q, k, -4.0)),
        ("first 16 keys far above the rest", 128, 512, lambda q, k: _shift_first(q, k, 4.0)),
        ("exp2 overflow against the old maximum (|s c| ~ 400)", 128, 320, lambda q, k: _spike(q, k, [100, 300], 40.0)),
    ):
        b, h = 2, 2
        C = 64 * h
        q2 = rnd(b * S, C, scale=1.0, seed=1234)
        kv = rnd(b * Sk, 2 * C, scale=1.0, seed=4321)
        build(q2.view(b, S, h, 64), kv[:, :C].view(b, Sk, h, 64))
        q = q2.view(b, S, h, 64).transpose(1, 2)
        k = kv[:, :C].reshape(b, Sk, h, 64).transpose(1, 2)
        v = kv[:, C:].reshape(b, Sk, h, 64).transpose(1, 2)
        ref = _sdpa(q, k, v).transpose(1, 2).reshape(b * S, C)
        o = torch.zeros(b * S, C, dtype=torch.float16, device=DEV)
        ops.attention(q2, kv[:, :C], kv[:, C:], o, batch=b, heads=h, Sq=S, Sk=Sk, inner=1, q_strides=(S, 0, 1),
                      kv_strides=(Sk, 0, 1))
        out.append(_res(f"attn[flash] forced rescale: {name}", o, ref, KTOL))
        if S == Sk:  # the shared-softmax kernel on the same Q / K (three V streams)
            b3 = 3
            qkv3 = rnd(b3 * S, 3 * C, scale=1.0, seed=99)
            qkv3[:S, :C] = q2[:S]
            qkv3[:S, C:2 * C] = kv[:S, :C]
            o3 = torch.zeros(b3 * S, C, dtype=torch.float16, device=DEV)
            ops.attention(qkv3[:, :C], qkv3[:, C:2 * C], qkv3[:, 2 * C:], o3, batch=b3, heads=h, Sq=S, Sk=S, inner=1,
                          q_strides=(S, 0, 1), kv_strides=(S, 0, 1), qk_mod=1)
            q3, k3, v3 = (qkv3[:, i * C:(i + 1) * C].view(b3, S, h, 64).transpose(1, 2).clone() for i in range(3))
            q3[1], k3[1], q3[2], k3[2] = q3[0], k3[0], q3[0], k3[0]
            out.append(_res(f"attn[shared softmax] forced rescale: {name}", o3,
                            _sdpa(q3, k3, v3).transpose(1, 2).reshape(b3 * S, C), KTOL))
    return out


def _spike(q, k, keys, gain):
    """k[:, key] := gain * (a query row's direction): that row's score at `key` dwarfs the others (in place, views of fp16)."""
    for n, key in enumerate(keys):
        rows = q[:, (7 + 13 * n) % q.shape[1]]            # [b, h, 64]: a different query row per spike
        k[:, key] = (gain * rows.float()).half()


def _shift_first(q, k, gain):
    """The first 16 keys get scores far below (gain < 0) / above (gain > 0) all later ones for EVERY query: both q and the
    first keys receive a large common component."""
    q[..., 0] = 6.0
    k[:, :16, :, 0] = gain * 6.0


def check_attention(naive_too=True):
    out = []
    for naive in ((False, True) if naive_too else (False,)):
        tag = "naive" if naive else "flash"
        # spatial self-attention out of a fused QKV buffer
        for (b, h, S) in [(3, 2, 200), (2, 5, 1024), (6, 1, 64), (1, 2, 4096)]:
            C = 64 * h
            qkv = rnd(b * S, 3 * C, scale=1.0)
            o = torch.zeros(b * S, C, dtype=torch.float16, device=DEV)
            ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=b, heads=h, Sq=S, Sk=S, inner=1,
                          q_strides=(S, 0, 1), kv_strides=(S, 0, 1), naive=naive)
            q, k, v = (qkv[:, i * C:(i + 1) * C].view(b, S, h, 64).transpose(1, 2) for i in range(3))
            ref = _sdpa(q, k, v).transpose(1, 2).reshape(b * S, C)
            out.append(_res(f"attn[{tag}] spatial b{b} h{h} S{S}", o, ref, KTOL))
        # PnP aliasing: Q,K of branches 1,2 come from branch 0
        b, h, S = 6, 2, 192
        C = 64 * h
        qkv = rnd(b * S, 3 * C)
        o = torch.zeros(b * S, C, dtype=torch.float16, device=DEV)
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=b, heads=h, Sq=S, Sk=S, inner=1,
                      q_strides=(S, 0, 1), kv_strides=(S, 0, 1), qk_mod=b // 3, naive=naive)
        q, k, v = (qkv[:, i * C:(i + 1) * C].view(b, S, h, 64).transpose(1, 2).clone() for i in range(3))
        c3 = b // 3
        q[c3:2 * c3], k[c3:2 * c3], q[2 * c3:], k[2 * c3:] = q[:c3], k[:c3], q[:c3], k[:c3]  # pnp_utils.py:192-196
        out.append(_res(f"attn[{tag}] spatial PnP q/k injection", o, _sdpa(q, k, v).transpose(1, 2).reshape(b * S, C), KTOL))
        # shared-softmax PnP kernel (one S/P per source element, three V streams): ragged and multi-tile shapes, and
        # agreement with the aliasing form of the same launch (flag bit3), which runs the plain kernel per branch
        for (b, h, S) in [(3, 1, 333), (6, 5, 1024), (48, 1, 130)]:
            C = 64 * h
            qkv = rnd(b * S, 3 * C, scale=1.0)
            o = torch.zeros(b * S, C, dtype=torch.float16, device=DEV)
            kw = dict(batch=b, heads=h, Sq=S, Sk=S, inner=1, q_strides=(S, 0, 1), kv_strides=(S, 0, 1), qk_mod=b // 3)
            ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, naive=naive, **kw)
            q, k, v = (qkv[:, i * C:(i + 1) * C].view(b, S, h, 64).transpose(1, 2).clone() for i in range(3))
            c3 = b // 3
            q[c3:2 * c3], k[c3:2 * c3], q[2 * c3:], k[2 * c3:] = q[:c3], k[:c3], q[:c3], k[:c3]
            out.append(_res(f"attn[{tag}] PnP shared-softmax b{b} h{h} S{S}", o,
                            _sdpa(q, k, v).transpose(1, 2).reshape(b * S, C), KTOL))
            if not naive:
                o2 = torch.zeros_like(o)
                saved, ops.ATTN_FLAGS = ops.ATTN_FLAGS, ops.ATTN_FLAGS | 8
                try:
                    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o2, **kw)
                finally:
                    ops.ATTN_FLAGS = saved
                out.append(_res(f"attn PnP shared-softmax == aliasing form b{b} h{h} S{S}", o, o2.float(), 1e-6))
        if not naive:
            # the 64x64-level launch itself (48 x 5 x 4096: the 8-wave / 3-stage form of the shared-softmax kernel): against the
            # aliasing form (plain kernel per branch) everywhere, against fp32 SDPA on one (source element, head)
            b, h, S = 48, 5, 4096
            C = 64 * h
            qkv = rnd(b * S, 3 * C, scale=1.0, seed=77)
            o = torch.zeros(b * S, C, dtype=torch.float16, device=DEV)
            kw = dict(batch=b, heads=h, Sq=S, Sk=S, inner=1, q_strides=(S, 0, 1), kv_strides=(S, 0, 1), qk_mod=b // 3)
            ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, **kw)
            o2 = torch.zeros_like(o)
            saved, ops.ATTN_FLAGS = ops.ATTN_FLAGS, ops.ATTN_FLAGS | 8
            try:
                ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o2, **kw)
            finally:
                ops.ATTN_FLAGS = saved
            out.append(_res(f"attn PnP shared-softmax == aliasing form b{b} h{h} S{S} (8-wave blocks)", o, o2.float(), 1e-6))
            # ... and with a ragged last block / last tile on both 8-wave kernels (2100 = 8 x 256 + 52 queries, 32 x 64 + 52 keys)
            br, hr, Sr = 96, 4, 2100
            Cr = 64 * hr
            qkvr = rnd(br * Sr, 3 * Cr, scale=1.0, seed=78)
            kwr = dict(batch=br, heads=hr, Sq=Sr, Sk=Sr, inner=1, q_strides=(Sr, 0, 1), kv_strides=(Sr, 0, 1), qk_mod=br // 3)
            o_r = torch.zeros(br * Sr, Cr, dtype=torch.float16, device=DEV)
            ops.attention(qkvr[:, :Cr], qkvr[:, Cr:2 * Cr], qkvr[:, 2 * Cr:], o_r, **kwr)
            o_r2 = torch.zeros_like(o_r)
            saved, ops.ATTN_FLAGS = ops.ATTN_FLAGS, ops.ATTN_FLAGS | 8
            try:
                ops.attention(qkvr[:, :Cr], qkvr[:, Cr:2 * Cr], qkvr[:, 2 * Cr:], o_r2, **kwr)
            finally:
                ops.ATTN_FLAGS = saved
            out.append(_res(f"attn PnP shared-softmax == aliasing form b{br} h{hr} S{Sr} (8-wave blocks, ragged)", o_r, o_r2.float(), 1e-6))
            xr = qkvr.view(br, Sr, 3, hr, 64)
            out.append(_res(f"attn PnP shared-softmax b{br} h{hr} S{Sr}, element 7 head 1 branch 2 vs fp32 SDPA",
                            o_r.view(br, Sr, hr, 64)[7 + 2 * (br // 3), :, 1],
                            _sdpa(xr[7, :, 0, 1][None, None], xr[7, :, 1, 1][None, None], xr[7 + 2 * (br // 3), :, 2, 1][None, None])[0, 0], KTOL))
            del qkvr, o_r, o_r2
            i0, hh = 5, 3
            x = qkv.view(b, S, 3, h, 64)
            q1, k1 = x[i0, :, 0, hh][None, None], x[i0, :, 1, hh][None, None]
            for br in range(3):
                v1 = x[i0 + br * (b // 3), :, 2, hh][None, None]
                got = o.view(b, S, h, 64)[i0 + br * (b // 3), :, hh]
                out.append(_res(f"attn PnP shared-softmax b{b} h{h} S{S}, element {i0} head {hh} branch {br} vs fp32 SDPA", got, _sdpa(q1, k1, v1)[0, 0], KTOL))
        # cross-attention: Sk=145, K/V shared by the F frames of a clip
        B_, Fr, S, h, Sk = 2, 3, 100, 2, 145
        C = 64 * h
        q2 = rnd(B_ * Fr * S, C)
        kv = rnd(B_ * Sk, 2 * C + 64)  # wider buffer: K at cols [0,C), V at [C,2C)
        o = torch.zeros(B_ * Fr * S, C, dtype=torch.float16, device=DEV)
        ops.attention(q2, kv[:, :C], kv[:, C:2 * C], o, batch=B_ * Fr, heads=h, Sq=S, Sk=Sk, inner=1,
                      q_strides=(S, 0, 1), kv_strides=(Sk, 0, 1), kv_div=Fr, naive=naive)
        q = q2.view(B_ * Fr, S, h, 64).transpose(1, 2)
        k = kv[:, :C].reshape(B_, Sk, h, 64).transpose(1, 2).repeat_interleave(Fr, 0)
        v = kv[:, C:2 * C].reshape(B_, Sk, h, 64).transpose(1, 2).repeat_interleave(Fr, 0)
        out.append(_res(f"attn[{tag}] cross Sk=145 kv_div", o, _sdpa(q, k, v).transpose(1, 2).reshape(-1, C), KTOL))
        # temporal: sequences stride by HW rows inside the [(b f) hw, C] matrix; with and without PnP
        for Fr in (16, 8, 40):
            for inj in (False, True):
                B_, HW, h = 3, 20, 2
                C = 64 * h
                qkv = rnd(B_ * Fr * HW, 3 * C)
                o = torch.zeros(B_ * Fr * HW, C, dtype=torch.float16, device=DEV)
                st = (Fr * HW, 1, HW)
                ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=B_ * HW, heads=h, Sq=Fr, Sk=Fr,
                              inner=HW, q_strides=st, kv_strides=st, qk_mod=HW if inj else 0, naive=naive)
                # reference layout of pnp_utils.py: [(B HW), F, C]
                def seq(x):
                    return x.reshape(B_, Fr, HW, h, 64).permute(0, 2, 3, 1, 4).reshape(B_ * HW, h, Fr, 64).clone()
                q, k, v = (seq(qkv[:, i * C:(i + 1) * C]) for i in range(3))
                if inj:
                    c3 = HW
                    q[c3:2 * c3], k[c3:2 * c3], q[2 * c3:], k[2 * c3:] = q[:c3], k[:c3], q[:c3], k[:c3]
                ref = _sdpa(q, k, v).reshape(B_, HW, h, Fr, 64).permute(0, 3, 1, 2, 4).reshape(B_ * Fr * HW, C)
                out.append(_res(f"attn[{tag}] temporal F{Fr} inject={inj}", o, ref, KTOL))
                if inj and not naive:   # the shared-softmax forms (one wave / block per source sequence) == per-branch aliasing, bit for bit
                    o2 = torch.zeros_like(o)
                    saved, ops.ATTN_FLAGS = ops.ATTN_FLAGS, ops.ATTN_FLAGS | 8
                    try:
                        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o2, batch=B_ * HW, heads=h, Sq=Fr, Sk=Fr,
                                      inner=HW, q_strides=st, kv_strides=st, qk_mod=HW)
                    finally:
                        ops.ATTN_FLAGS = saved
                    out.append(_res(f"attn temporal F{Fr} PnP shared-softmax == aliasing form", o, o2.float(), 1e-6 if Fr > 16 else 0.0))
    # deferred rescale (the running maximum only advances when a probability would exceed 2^8): wide score ranges, and
    # keys whose scores grow along the sequence so that the maximum keeps jumping by more than the threshold
    for name, qs, ramp in (("wide scores (|s c| up to ~60)", 3.0, False), ("ramped keys (max jumps every tile)", 1.0, True)):
        b, h, S = 2, 2, 1024
        C = 64 * h
        qkv = rnd(b * S, 3 * C, scale=qs, seed=4242)
        if ramp:
            grow = torch.linspace(0.0, 12.0, S, device=DEV).repeat(b).unsqueeze(1).half()
            qkv[:, C:2 * C] = (qkv[:, C:2 * C].float() * (1.0 + grow.float())).half()
        o = torch.zeros(b * S, C, dtype=torch.float16, device=DEV)
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=b, heads=h, Sq=S, Sk=S, inner=1,
                      q_strides=(S, 0, 1), kv_strides=(S, 0, 1))
        q, k, v = (qkv[:, i * C:(i + 1) * C].view(b, S, h, 64).transpose(1, 2) for i in range(3))
        out.append(_res(f"attn[flash] {name}", o, _sdpa(q, k, v).transpose(1, 2).reshape(b * S, C), KTOL))
        # the shared-softmax kernel on the same data (three V streams, branch 0's Q / K)
        b3 = 3
        qkv3 = rnd(b3 * S, 3 * C, scale=qs, seed=777)
        if ramp:
            grow3 = torch.linspace(0.0, 12.0, S, device=DEV).repeat(b3).unsqueeze(1)
            qkv3[:, C:2 * C] = (qkv3[:, C:2 * C].float() * (1.0 + grow3)).half()
        o3 = torch.zeros(b3 * S, C, dtype=torch.float16, device=DEV)
        ops.attention(qkv3[:, :C], qkv3[:, C:2 * C], qkv3[:, 2 * C:], o3, batch=b3, heads=h, Sq=S, Sk=S, inner=1,
                      q_strides=(S, 0, 1), kv_strides=(S, 0, 1), qk_mod=1)
        q, k, v = (qkv3[:, i * C:(i + 1) * C].view(b3, S, h, 64).transpose(1, 2).clone() for i in range(3))
        q[1], k[1], q[2], k[2] = q[0], k[0], q[0], k[0]
        out.append(_res(f"attn[shared softmax] {name}", o3, _sdpa(q, k, v).transpose(1, 2).reshape(b3 * S, C), KTOL))
    out += check_attention_forced_rescale()
    # 8-wave (256-query) blocks: taken for launches with >= 1024 such blocks and Sq, Sk >= 1024; the same launch forced onto
    # 4-wave blocks (flag bit2) must give the same bits (a wave's instruction stream does not depend on the block shape)
    b, h, S = 13, 5, 4096
    C = 64 * h
    qkv = rnd(b * S, 3 * C, scale=1.0, seed=2024)
    o8 = torch.zeros(b * S, C, dtype=torch.float16, device=DEV)
    o4 = torch.zeros_like(o8)
    kw8 = dict(batch=b, heads=h, Sq=S, Sk=S, inner=1, q_strides=(S, 0, 1), kv_strides=(S, 0, 1))
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o8, **kw8)
    saved, ops.ATTN_FLAGS = ops.ATTN_FLAGS, ops.ATTN_FLAGS | 4
    try:
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o4, **kw8)
    finally:
        ops.ATTN_FLAGS = saved
    out.append(_res("attn[flash] 8-wave blocks == 4-wave blocks (b13 h5 S4096)", o8, o4.float(), 1e-6))
    q, k, v = (qkv[:2 * S, i * C:(i + 1) * C].view(2, S, h, 64).transpose(1, 2) for i in range(3))
    out.append(_res("attn[flash] 8-wave blocks vs fp32 SDPA (first 2 of b13 h5 S4096)", o8[:2 * S],
                    _sdpa(q, k, v).transpose(1, 2).reshape(2 * S, C), KTOL))
    B_, HW, h, Fr = 2, 12, 2, 128  # temporal sequences of 128 frames (BASELINE config 5): frame stride HW
    C = 64 * h
    qkv = rnd(B_ * Fr * HW, 3 * C)
    st = (Fr * HW, 1, HW)
    o = torch.zeros(B_ * Fr * HW, C, dtype=torch.float16, device=DEV)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=B_ * HW, heads=h, Sq=Fr, Sk=Fr, inner=HW, q_strides=st,
                  kv_strides=st)
    def seq128(x):
        return x.reshape(B_, Fr, HW, h, 64).permute(0, 2, 3, 1, 4).reshape(B_ * HW, h, Fr, 64)
    q, k, v = (seq128(qkv[:, i * C:(i + 1) * C]) for i in range(3))
    ref = _sdpa(q, k, v).reshape(B_, HW, h, Fr, 64).permute(0, 3, 1, 2, 4).reshape(B_ * Fr * HW, C)
    out.append(_res("attn[flash] temporal F128 (frame stride)", o, ref, KTOL))
    # every KV-loop length 1 .. 7 tiles with ragged last tiles, shared K/V (kv_div), Q/K aliasing on the plain kernel
    for (b, h, S, Sk, kv_div, qk_mod) in [(2, 1, 20, 3, 1, 0), (2, 2, 130, 17, 1, 0), (1, 1, 33, 63, 1, 0),  # tiny ragged KV: whole steps masked
                                           (2, 1, 128, 64, 1, 0), (2, 2, 128, 128, 1, 0), (4, 1, 256, 145, 2, 0),
                                           (1, 2, 128, 200, 1, 0), (2, 1, 128, 320, 1, 0), (1, 1, 256, 384, 1, 0),
                                           (4, 2, 128, 448, 1, 2), (1, 5, 4096, 4096, 1, 0)]:
        C = 64 * h
        q2 = rnd(b * S, C, scale=1.0)
        kv = rnd((b // kv_div) * Sk, 2 * C, scale=1.0)
        o = torch.zeros(b * S, C, dtype=torch.float16, device=DEV)
        ops.attention(q2, kv[:, :C], kv[:, C:], o, batch=b, heads=h, Sq=S, Sk=Sk, inner=1, q_strides=(S, 0, 1),
                      kv_strides=(Sk, 0, 1), kv_div=kv_div, qk_mod=qk_mod)
        q = q2.view(b, S, h, 64).transpose(1, 2).clone()
        k = kv[:, :C].reshape(b // kv_div, Sk, h, 64).transpose(1, 2).repeat_interleave(kv_div, 0).clone()
        v = kv[:, C:].reshape(b // kv_div, Sk, h, 64).transpose(1, 2).repeat_interleave(kv_div, 0)
        if qk_mod:
            for i in range(b):
                q[i], k[i] = q[i % qk_mod], k[i % qk_mod]
        out.append(_res(f"attn[flash] b{b} h{h} S{S} Sk{Sk} kv_div{kv_div} qk_mod{qk_mod}", o,
                        _sdpa(q, k, v).transpose(1, 2).reshape(b * S, C), KTOL))
    # small-head generic kernel (image_latents_temporal_encoder: 2 heads x dim 4)
    B_, Fr, HW, h, d = 2, 6, 10, 2, 4
    qkv = rnd(B_ * Fr * HW, 3 * h * d)
    C = h * d
    o = torch.zeros(B_ * Fr * HW, C, dtype=torch.float16, device=DEV)
    st = (Fr * HW, 1, HW)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=B_ * HW, heads=h, Sq=Fr, Sk=Fr, inner=HW,
                  q_strides=st, kv_strides=st, scale=d ** -0.5, head_dim=d)
    def seq(x):
        return x.reshape(B_, Fr, HW, h, d).permute(0, 2, 3, 1, 4).reshape(B_ * HW, h, Fr, d)
    q, k, v = (seq(qkv[:, i * C:(i + 1) * C]) for i in range(3))
    ref = _sdpa(q, k, v).reshape(B_, HW, h, Fr, d).permute(0, 3, 1, 2, 4).reshape(-1, C)
    out.append(_res("attn small head_dim 4", o, ref, KTOL))
    return out



# ------------------------------------------------------------------------------------------------ CLIP towers (F1)
def check_attention_small_mfma():
    """Whole-sequence MFMA attention of the CLIP towers (``anyv2v_attention_small_f16``: head_dim a multiple of 8 up to 160, Sk <= 288) vs
    fp32 SDPA and vs the one-thread-per-query kernel it replaces: ViT-H/14 (257 tokens, 16 x 80), the text tower (77 tokens, 16 x 64,
    causal), head_dim 128, query counts that are not multiples of the 64-query block, column windows of a fused QKV matrix."""
    out = []
    # (head_dim 40 / 160: ConsistI2V's temporal attention at C = 320 / 1280 with 8 heads -- not multiples of 16 / five 32-column groups)
    for (B, h, S, d, causal) in [(2, 16, 257, 80, False), (3, 16, 77, 64, True), (1, 4, 200, 128, False), (2, 3, 50, 96, True),
                                 (1, 2, 288, 64, False), (3, 8, 24, 40, False), (2, 8, 77, 160, False), (5, 8, 16, 160, True),
                                 (2, 4, 90, 56, False),
                                 # longer than 288 keys (or 96 at head_dim 160): the 96-key loop kernel with the online softmax -- SEINE's
                                 # spatial attention (8 heads x 40 / 80 / 160 over 2560 / 640 / 160 tokens)
                                 (1, 8, 2560, 40, False), (2, 8, 640, 80, False), (3, 8, 160, 160, False), (1, 2, 401, 56, True),
                                 (2, 3, 289, 24, False), (1, 4, 1000, 128, True)]:
        H = h * d
        qkv = rnd(B * S, 3 * H, seed=S + d)
        o = torch.zeros(B * S, H, dtype=torch.float16, device=DEV)
        kw = dict(batch=B, heads=h, Sq=S, Sk=S, inner=1, q_strides=(S, 0, 1), kv_strides=(S, 0, 1), scale=d ** -0.5, head_dim=d,
                  causal=causal)
        ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], o, **kw)
        sp = lambda x: x.float().view(B, S, h, d).transpose(1, 2)
        ref = F.scaled_dot_product_attention(sp(qkv[:, :H]), sp(qkv[:, H:2 * H]), sp(qkv[:, 2 * H:]), is_causal=causal)
        ref = ref.transpose(1, 2).reshape(B * S, H)
        out.append(_res(f"attention[small mfma] B{B} h{h} S{S} d{d} causal={causal} vs fp32 SDPA", o, ref, KTOL))
        o2 = torch.zeros_like(o)
        ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], o2, naive=True, **kw)
        out.append(_res(f"attention[small mfma] == one-thread-per-query kernel S{S} d{d}", o, o2.float(), 1.5e-3))
    # the temporal layout of the ConsistI2V blocks: one sequence per (clip, pixel), its F rows HW apart; keys = F frames + 8 gathered rows
    Bc, Fr, HW, h, d = 2, 16, 48, 8, 40
    C = h * d
    q = rnd(Bc * Fr * HW, C, seed=5)
    kv = rnd(Bc * HW * (Fr + 8), 2 * C, seed=6)
    o = torch.zeros(Bc * Fr * HW, C, dtype=torch.float16, device=DEV)
    kw = dict(batch=Bc * HW, heads=h, Sq=Fr, Sk=Fr + 8, inner=HW, q_strides=(Fr * HW, 1, HW), kv_strides=(HW * (Fr + 8), Fr + 8, 1),
              scale=d ** -0.5, head_dim=d)
    ops.attention(q, kv[:, :C], kv[:, C:], o, **kw)
    qs = q.float().view(Bc, Fr, HW, h, d).permute(0, 2, 3, 1, 4)                      # [B, HW, h, F, d]
    ks = kv[:, :C].float().view(Bc, HW, Fr + 8, h, d).permute(0, 1, 3, 2, 4)
    vs = kv[:, C:].float().view(Bc, HW, Fr + 8, h, d).permute(0, 1, 3, 2, 4)
    ref = F.scaled_dot_product_attention(qs, ks, vs).permute(0, 3, 1, 2, 4).reshape(Bc * Fr * HW, C)
    out.append(_res("attention[small mfma] temporal layout (inner = HW), 8 x 40, 16 queries x 24 keys vs fp32 SDPA", o, ref, KTOL))
    return out


def check_clip():
    """``anyv2v_amd.clip`` on the real kernels vs ``transformers``' CLIPTextModel / CLIPVisionModelWithProjection in
    fp32 on the CPU, same weights: a tiny pair (3 / 2 layers) and the checkpoint's widths (text 1024 / 16 heads x 64 / 4096 with
    77 tokens and the causal mask; vision ViT-H/14: 1280 / 16 heads x 80 / 5120, 257 tokens, projection 1024) at 2 layers."""
    import transformers
    from anyv2v_amd.clip import CLIPTextTower, CLIPTowerConfig, CLIPVisionTower
    out = []
    for tag, tk, vk in (
        ("tiny", dict(vocab_size=100, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                      max_position_embeddings=16),
         dict(hidden_size=160, intermediate_size=320, num_hidden_layers=2, num_attention_heads=2, image_size=224, patch_size=32,
              projection_dim=24)),
        ("checkpoint widths, 2 layers", dict(vocab_size=1000, hidden_size=1024, intermediate_size=4096, num_hidden_layers=2,
                                             num_attention_heads=16, max_position_embeddings=77),
         dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=2, num_attention_heads=16, image_size=224, patch_size=14,
              projection_dim=1024)),
    ):
        torch.manual_seed(7)
        tcfg = transformers.CLIPTextConfig(bos_token_id=1, eos_token_id=2, hidden_act="gelu", **tk)
        vcfg = transformers.CLIPVisionConfig(hidden_act="gelu", **vk)
        tm, vm = transformers.CLIPTextModel(tcfg).eval(), transformers.CLIPVisionModelWithProjection(vcfg).eval()
        with torch.no_grad():
            for m in (tm, vm):
                for n_, p_ in m.named_parameters():
                    if p_.dim() >= 2:
                        p_.normal_(0, 0.6 / p_.shape[-1] ** 0.5 if p_.dim() == 2 else 0.02)
                    elif "bias" in n_:
                        p_.normal_(0, 0.05)
            # the comparison is against fp32 on the weights the towers hold in fp16
            for m in (tm, vm):
                for p_ in m.parameters():
                    p_.copy_(p_.half().float())
        S = tk["max_position_embeddings"]
        ids = torch.randint(3, tk["vocab_size"], (3, S))
        ids[:, -2:] = 2
        text = CLIPTextTower(CLIPTowerConfig.from_hf(tcfg.to_dict()), tm.state_dict()).to(DEV)
        with torch.no_grad():
            want = tm(ids, output_hidden_states=True)
        ln = getattr(tm, "text_model", tm).final_layer_norm
        hs = text.hidden_states(ids)
        out.append(_res(f"clip text [{tag}] last hidden state (pre-LN, causal)", hs[-1], want.hidden_states[-1], 6e-3))
        for skip in (None, 1):
            with torch.no_grad():
                ref = want.last_hidden_state if skip is None else ln(want.hidden_states[-(skip + 1)])
            out.append(_res(f"clip text [{tag}] encode clip_skip={skip}", text.encode_ids(ids, skip), ref, 8e-3))
        vis = CLIPVisionTower(CLIPTowerConfig.from_hf(vcfg.to_dict()), vm.state_dict()).to(DEV)
        px = torch.randn(2, 3, 224, 224)
        with torch.no_grad():
            ref = vm(pixel_values=px.half().float()).image_embeds
        out.append(_res(f"clip vision [{tag}] image_embeds", vis.image_embeds(px), ref, 8e-3))
    return out

# ------------------------------------------------------------------------------------------------ erf-GELU, every fp16 input
def check_gelu_all_inputs():
    """The GEMM epilogues' erf-GELU (``av_gelu`` in csrc/common.h: max(x,0) - |x| q with the Abramowitz-Stegun 7.1.26 tail) on
    EVERY finite fp16 input, through the GELU and the GEGLU epilogue of the tile kernel and the GEGLU epilogue of the
    weight-stationary one: at most 1 ulp from the correctly rounded exact GELU (float64 erf), and that on < 1 % of the inputs
    (the reference's activation is ``F.gelu`` with the exact erf, diffusers GEGLU; VERDICT r2 #4: no change of semantics)."""
    out = []
    bits = torch.arange(0, 65536, dtype=torch.int32, device=DEV).to(torch.int16)
    x = bits.view(torch.float16)
    x = x[torch.isfinite(x)]                              # 63 488 values
    n = x.numel()
    exact = (0.5 * x.double() * (1.0 + torch.erf(x.double() / math.sqrt(2.0))))

    def ulps(y):   # distance in fp16 steps from the correctly rounded result (monotone integer key of an fp16 value)
        def key(h):
            i = h.view(torch.int16).to(torch.int32)
            return torch.where(i < 0, -(i & 0x7FFF), i)
        return (key(y) - key(exact.to(torch.float16))).abs()

    def report(tag, y):
        d = ulps(y)
        frac = float((d > 0).float().mean())
        worst = int(d.max())
        out.append(dict(name=f"erf-GELU over all {n} finite fp16 inputs, {tag}: {100 * frac:.2f} % differ, worst {worst} ulp",
                        err=float(worst), tol=1.0, ok=worst <= 1 and frac < 0.01))

    for K, M_pad in ((64, 0), (320, 32768)):   # K = 320 with >= 32768 rows: the weight-stationary kernel
        M = max(n, M_pad)
        a = torch.zeros(M, K, dtype=torch.float16, device=DEV)
        a[:n, 0] = x
        if K == 64:
            w = torch.zeros(128, K, dtype=torch.float16, device=DEV)
            w[:, 0] = 1.0
            report("GELU epilogue (tile kernel)", ops.gemm(a, w, act=ops.ACT_GELU)[:n, 5])
        # GEGLU: value column = 1 (through the bias), gate column = x.  Packed layout: groups of [16 value | 16 gate] columns
        N = 320 if K == 320 else 128
        w = torch.zeros(N, K, dtype=torch.float16, device=DEV)
        b = torch.zeros(N, dtype=torch.float16, device=DEV)
        for g in range(N // 32):
            w[g * 32 + 16:g * 32 + 32, 0] = 1.0
            b[g * 32:g * 32 + 16] = 1.0
        y = ops.gemm(a, w, bias=b, act=ops.ACT_GEGLU)
        report("GEGLU epilogue (%s)" % ("weight-stationary kernel" if K == 320 else "tile kernel"), y[:n, 3])
    return out


# ------------------------------------------------------------------------------------------------ elementwise
def check_elementwise():
    out = []
    x = rnd(1000, 33)
    out.append(_res("silu", ops.silu(x), F.silu(x.float()), 2e-3))
    a, b = rnd(777, 13), rnd(777, 13)
    out.append(_res("add", ops.add(a, b), a.float() + b.float(), 2e-3))
    t = torch.tensor([981.0, 1.0, 500.0], device=DEV)
    e = ops.timestep_embedding(t, 320)
    half = 160
    fr = torch.exp(-math.log(10000.0) * torch.arange(half, device=DEV).float() / half)
    arg = t[:, None] * fr[None]
    out.append(_res("timestep embedding", e, torch.cat([arg.cos(), arg.sin()], -1), 2e-3))
    B, C, Fr, H, W = 2, 4, 3, 5, 6
    lat = rnd(B, C, Fr, H, W)
    tok = torch.zeros(B * Fr * H * W, 16, dtype=torch.float16, device=DEV)
    ops.ncfhw_to_tokens(lat, tok, col0=4)
    ref = lat.permute(0, 2, 3, 4, 1).reshape(-1, C)
    out.append(_res("ncfhw_to_tokens", tok[:, 4:8], ref, 0.0))
    back = ops.tokens_to_ncfhw(tok, B, C, Fr, H, W, col0=4)
    out.append(_res("tokens_to_ncfhw", back, lat, 0.0))
    xp = rnd(2 * 8 * 8, 32)
    y = ops.adaptive_avgpool(xp, 2, 8, 8, 32, 32)
    ref = F.adaptive_avg_pool2d(xp.float().view(2, 8, 8, 32).permute(0, 3, 1, 2), (32, 32)).permute(0, 2, 3, 1).reshape(-1, 32)
    out.append(_res("adaptive_avgpool 8->32", y, ref, 2e-3))
    xp = rnd(2 * 64 * 64, 8)
    y = ops.adaptive_avgpool(xp, 2, 64, 64, 32, 32)
    ref = F.adaptive_avg_pool2d(xp.float().view(2, 64, 64, 8).permute(0, 3, 1, 2), (32, 32)).permute(0, 2, 3, 1).reshape(-1, 8)
    out.append(_res("adaptive_avgpool 64->32", y, ref, 2e-3))
    # fused CFG + DDIM step
    Fr, H, W = 3, 4, 5
    vt = rnd(3 * Fr * H * W, 8)
    latx = rnd(1, 4, Fr, H, W)
    coef = torch.tensor([0.8, 0.6, 0.9, math.sqrt(1 - 0.81)], dtype=torch.float32, device=DEV)
    o = torch.zeros_like(latx)
    ops.cfg_ddim_step(vt, 1, 2, 9.0, coef, latx, o)
    v3 = vt[:, :4].float().view(3, Fr, H * W, 4).permute(0, 3, 1, 2).reshape(3, 4, Fr, H, W)
    v = v3[1] + 9.0 * (v3[2] - v3[1])
    xx = latx.float()[0]
    x0 = 0.8 * xx - 0.6 * v
    eps = 0.8 * v + 0.6 * xx
    out.append(_res("cfg + ddim step", o[0], 0.9 * x0 + float(coef[3]) * eps, KTOL))
    o2 = ops.ddim_step(v3[2].half(), latx[0], 0.8, 0.6, 0.9, float(coef[3]))
    x0 = 0.8 * xx - 0.6 * v3[2]
    eps = 0.8 * v3[2] + 0.6 * xx
    out.append(_res("ddim step", o2, 0.9 * x0 + float(coef[3]) * eps, 3e-3))
    # row gather with column windows (ConsistI2V's [own frame ; first frame] / [frames ; first-frame window] key sequences)
    src = rnd(300, 96)
    idx = torch.randint(0, 300, (517,), device=DEV, dtype=torch.int32)
    dst = torch.zeros(517, 80, dtype=torch.float16, device=DEV)
    ops.gather_rows(src, 32, idx, dst, 16, 64)
    want = torch.zeros_like(dst)
    want[:, 16:80] = src[idx.long(), 32:96]
    out.append(dict(name="gather_rows (column windows)", err=float((dst.float() - want.float()).abs().max()), tol=0.0,
                    ok=bool(torch.equal(dst, want))))
    # rotary position embedding in place vs the formula of rotary_embedding.py:29-49 in fp32
    HW_, F_, C_ = 6, 5, 96
    xr = rnd(2 * F_ * HW_, C_)
    got = ops.rotary(xr.clone(), 32, 48, HW_, F_)
    pos = ((torch.arange(xr.shape[0], device=DEV) // HW_) % F_).float()
    freq = 10000.0 ** (-torch.arange(0, 48, 2, device=DEV).float() / 48)
    ang = (pos[:, None] * freq[None]).repeat_interleave(2, dim=1)
    t = xr[:, 32:80].float()
    rot = torch.stack([-t[:, 1::2], t[:, 0::2]], -1).reshape(t.shape)
    want = xr.float().clone()
    want[:, 32:80] = t * ang.cos() + rot * ang.sin()
    out.append(_res("rotary (window of columns, frame positions)", got, want, 2e-3))
    out.append(dict(name="rotary leaves the other columns alone", err=0.0, tol=0.0,
                    ok=bool(torch.equal(got[:, :32], xr[:, :32]) and torch.equal(got[:, 80:], xr[:, 80:]))))
    return out


# ------------------------------------------------------------------------------------------------ model level
def build_pair(cfg_name="mini", seed=1234, oracle_device="cpu"):
    from anyv2v_amd.unet import I2VGenXLUNet, I2VGenXLUNetConfig
    from oracle.unet_oracle import UNetConfig, build_oracle, random_state_dict
    ocfg = UNetConfig.mini() if cfg_name == "mini" else UNetConfig.i2vgen_xl()
    ncfg = I2VGenXLUNetConfig.mini() if cfg_name == "mini" else I2VGenXLUNetConfig()
    sd = random_state_dict(ocfg, seed)
    oracle = build_oracle(ocfg, sd, dtype=torch.float32, device=oracle_device)
    with torch.device("meta"):
        native = I2VGenXLUNet(ncfg)
    native = native.to_empty(device=DEV)
    native.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=True)
    return native, oracle, ocfg


def check_unet_golden():
    """Native UNet + native PnP hooks vs the fixture produced by the reference's own pnp_utils.py on the oracle.  On the GPU
    the bound is 2 x the error of the torch-eager fp16 oracle (with ``oracle.pnp_oracle`` hooks) against the same fixture."""
    import types
    from anyv2v_amd import pnp_utils
    from oracle import pnp_oracle
    out = []
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "pnp_hooks_mini.pt"))
    calibrate = DEV != "cpu"
    m = full_models("mini", gold["mini_seed"], want=("native",) + (("o16",) if calibrate else ()))
    native, o16 = m["native"], m.get("o16")
    inp = {k: v.to(DEV) for k, v in gold["inputs"].items()}
    kw = dict(fps=inp["fps"], image_latents=inp["image_latents"].half(), image_embeddings=inp["image_embeddings"].half(),
              encoder_hidden_states=inp["encoder_hidden_states"].half())
    sample = inp["sample"].half()

    def compare(name, t, ref):
        v = native(sample, t, **kw)[0]
        if calibrate:
            with torch.no_grad():
                v16 = o16(sample, t, **kw)[0]
            out.append(_calibrated(name, v, ref, v16))
        else:
            out.append(_res(name, v.cpu(), ref, 3e-2))

    pipe = types.SimpleNamespace(unet=native)
    pnp_utils.clear_time(pipe)
    compare("unet mini (no hooks) vs reference-run fixture", 981, gold["v_nohook_t981"])
    n, p = gold["n_steps"], gold["pnp"]
    pipe = _hook_all(m, ("o16",) if calibrate else (), n_steps=n, ratios=(p["pnp_f_t"], p["pnp_spatial_attn_t"], p["pnp_temp_attn_t"]))
    try:
        for t in (981, 701, 301, 101):
            pnp_utils.register_time(pipe, t)
            if calibrate:
                pnp_oracle.register_time(o16, t)
            compare(f"unet mini + native PnP hooks t={t} vs reference pnp_utils fixture", t, gold[f"v_hook_t{t}"])
    finally:
        _unhook_all(m, ("o16",) if calibrate else (), pipe)
    return out


def check_foreign_hooks(source="oracle"):
    """Seams B1 / B2: torch-style hook code from OUTSIDE the product package drives the native UNet.

    ``source="reference"``: the reference's own ``i2vgen-xl/pnp_utils.py`` (imported verbatim through
    ``oracle.ref_stubs``; only where /root/reference exists, i.e. the CPU suite) -- its ``register_conv_injection``
    replaces ``up_blocks[1].resnets[1].forward`` (``pnp_utils.py:130-131``) and its ``register_*_attention_pnp`` plug
    ``ModifiedSpaAttnProcessor`` / ``ModifiedTmpAttnProcessor`` objects into 16 ``attn1.processor`` slots (``:235-242,340-347``).
    ``source="oracle"``: ``oracle.pnp_oracle``'s restatement of the same hooks (pinned to the reference by the golden
    fixture) -- what the GPU box can run.  The result must equal the fixture produced by the reference hooks on the oracle
    UNet, and the native hooks' result on the same model."""
    import types
    from anyv2v_amd import pnp_utils
    from anyv2v_amd.unet import Attention, ResnetBlock2D
    out = []
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "pnp_hooks_mini.pt"))
    inp = {k: v.to(DEV) for k, v in gold["inputs"].items()}
    kw = dict(fps=inp["fps"], image_latents=inp["image_latents"].half(), image_embeddings=inp["image_embeddings"].half(),
              encoder_hidden_states=inp["encoder_hidden_states"].half())
    sample = inp["sample"].half()
    n, p = gold["n_steps"], gold["pnp"]
    ts = torch.arange(n).flip(0) * (1000 // n) + 1
    k = lambda name: ts[: int(n * p[name])]
    # native hooks first (reference result for "same model, same kernels")
    native, _, _ = build_pair("mini", gold["mini_seed"])
    pipe = types.SimpleNamespace(unet=native)
    pnp_utils.register_conv_injection(pipe, k("pnp_f_t"))
    pnp_utils.register_spatial_attention_pnp(pipe, k("pnp_spatial_attn_t"))
    pnp_utils.register_temp_attention_pnp(pipe, k("pnp_temp_attn_t"))
    v_native = {}
    for t in (981, 701, 301, 101):
        pnp_utils.register_time(pipe, t)
        v_native[t] = native(sample, t, **kw)[0].float().cpu()
    # foreign hooks on a fresh native model
    native, _, _ = build_pair("mini", gold["mini_seed"])
    pipe = types.SimpleNamespace(unet=native)
    calls = {"attn": 0, "resnet": 0}
    orig_a, orig_r = Attention._run_foreign, ResnetBlock2D._run_foreign

    def count_a(self, *a, **kk):
        calls["attn"] += 1
        return orig_a(self, *a, **kk)

    def count_r(self, *a, **kk):
        calls["resnet"] += 1
        return orig_r(self, *a, **kk)

    Attention._run_foreign, ResnetBlock2D._run_foreign = count_a, count_r
    try:
        if source == "reference":
            from oracle import ref_stubs
            ref = ref_stubs.load_reference_pnp_utils()
            ref.register_conv_injection(pipe, k("pnp_f_t"))
            ref.register_spatial_attention_pnp(pipe, k("pnp_spatial_attn_t"))
            ref.register_temp_attention_pnp(pipe, k("pnp_temp_attn_t"))
            reg_time = lambda t: ref.register_time(pipe, t)
        else:
            from oracle import pnp_oracle
            pnp_oracle.register_conv_injection(native, k("pnp_f_t"))
            pnp_oracle.register_spatial_attention_pnp(native, k("pnp_spatial_attn_t"))
            pnp_oracle.register_temp_attention_pnp(native, k("pnp_temp_attn_t"))
            reg_time = lambda t: pnp_oracle.register_time(native, t)
        assert pnp_utils.has_foreign_hooks(native)
        for t in (981, 701, 301, 101):
            reg_time(t)
            v = native(sample, t, **kw)[0].float().cpu()
            out.append(_res(f"native UNet driven by the {source} hook code t={t} vs reference pnp_utils fixture", v, gold[f"v_hook_t{t}"], 3e-2))
            out.append(_res(f"native UNet: {source} hook code == native hooks t={t}", v, v_native[t], 6e-3))
    finally:
        Attention._run_foreign, ResnetBlock2D._run_foreign = orig_a, orig_r
    # 16 foreign processors + 1 replaced ResNet forward, 4 forwards
    out.append(dict(name=f"seam B1: Attention._run_foreign ran 16 x 4 times ({calls['attn']})", err=float(calls["attn"] != 64), l2=0.0, tol=0.5,
                    ok=calls["attn"] == 64))
    out.append(dict(name=f"seam B2: ResnetBlock2D replaced forward ran 4 times ({calls['resnet']})", err=float(calls["resnet"] != 4), l2=0.0,
                    tol=0.5, ok=calls["resnet"] == 4))
    # the module's own torch-style forward (NCHW in / out) == its token-layout run
    blk = native.up_blocks[2].resnets[0]
    del_forward = native.up_blocks[1].resnets[1].__dict__.pop("forward", None)  # noqa: F841 (restore the class forward)
    g = torch.Generator().manual_seed(3)
    N_, H_ = 4, 8
    x = (torch.randn(N_, blk.in_channels, H_, H_, generator=g)).half().to(DEV)
    temb = torch.randn(N_, native.cfg.time_embed_dim, generator=g).half().to(DEV)
    y = blk(x, temb)
    xf, tf = x.float().cpu(), temb.float().cpu()
    P = {kk: vv.float().cpu() for kk, vv in blk.state_dict().items()}
    h = F.conv2d(F.silu(F.group_norm(xf, 32, P["norm1.weight"], P["norm1.bias"], 1e-5)), P["conv1.weight"], P["conv1.bias"], padding=1)
    h = h + F.linear(F.silu(tf), P["time_emb_proj.weight"], P["time_emb_proj.bias"])[:, :, None, None]
    h = F.conv2d(F.silu(F.group_norm(h, 32, P["norm2.weight"], P["norm2.bias"], 1e-5)), P["conv2.weight"], P["conv2.bias"], padding=1)
    ref_y = F.conv2d(xf, P["conv_shortcut.weight"], P["conv_shortcut.bias"]) + h
    out.append(_res("ResnetBlock2D torch-style forward(NCHW, temb) vs torch fp32", y.float().cpu(), ref_y, 4e-3))
    return out


def config1_inputs(cfg, B, Fr=8, hw=32, seed=8888):
    """BASELINE config 1 inputs: random latents, frame-position planes, seed 8888."""
    g = torch.Generator().manual_seed(seed)
    h, w = (hw, hw) if isinstance(hw, int) else hw
    sample = torch.randn(B, 4, Fr, h, w, generator=g)
    il = torch.randn(B, 4, Fr, h, w, generator=g)
    for i in range(1, Fr):
        il[:, :, i] = i / (Fr - 1)
    ehs = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    ie = torch.randn(B, 1, cfg.cross_attention_dim, generator=g)
    return dict(sample=sample, image_latents=il, encoder_hidden_states=ehs, image_embeddings=ie, fps=torch.tensor([8] * B))


def check_unet_vs_oracle(cfg_name="mini", B=3, Fr=4, hw=8, with_pnp=True, tol=3e-2, report=None, calibrate=True):
    """Single denoise step, HIP UNet (fp16) vs CPU oracle (fp32) on identical fp16-rounded weights and inputs.  On the GPU
    every comparison is bounded by 2 x the error of the same oracle model run by PyTorch-ROCm eager in fp16 on this GPU --
    the stand-in for "the reference's fp16 path" (SURVEY.md 8(c) tolerance policy); ``tol`` only applies to the CPU
    emulation of the ops (host-logic tests)."""
    import time
    import types
    from anyv2v_amd import pnp_utils
    from oracle import pnp_oracle
    out = []
    calibrate = calibrate and DEV != "cpu"
    m = full_models(cfg_name, 1234, want=("native", "ocpu") + (("o16",) if calibrate else ()))
    native, oracle, ocfg = m["native"], m["ocpu"], m["ocfg"]
    o16 = m.get("o16")
    inp = config1_inputs(ocfg, B, Fr, hw)
    inp16 = {k: (v.half() if v.is_floating_point() else v) for k, v in inp.items()}
    kw_o = _cond_kw(inp16, "cpu", torch.float32)
    kw_n = _cond_kw(inp16, DEV, torch.float16)

    def compare(name, t):
        with torch.no_grad():
            t0 = time.time()
            vo = oracle(inp16["sample"].float(), t, **kw_o)[0]
            t_cpu = time.time() - t0
            v16 = o16(inp16["sample"].to(DEV), t, **kw_n)[0] if calibrate else None
        vn = native(inp16["sample"].to(DEV), t, **kw_n)[0]
        _sync()
        out.append(_calibrated(name, vn, vo, v16) if calibrate else _res(name, vn.cpu(), vo, tol))
        return t_cpu

    hw_ = (hw, hw) if isinstance(hw, int) else tuple(hw)
    hw = f"{hw_[0]}x{hw_[1]}"
    t_cpu = compare(f"unet {cfg_name} B{B} F{Fr} {hw} step vs oracle", 981)
    if report is not None:
        report[f"cpu_oracle_seconds_{cfg_name}_B{B}"] = t_cpu
    if with_pnp and B == 3:
        ts = [981 - 20 * i for i in range(50)]
        pipe = _hook_all(m, ("ocpu",) + (("o16",) if calibrate else ()), ratios=(0.2, 0.5, 0.8))
        try:
            for t in (981, 301):
                pnp_utils.register_time(pipe, t)
                pnp_oracle.register_time(oracle, t)
                if calibrate:
                    pnp_oracle.register_time(o16, t)
                compare(f"unet {cfg_name} B3 F{Fr} {hw} PnP step t={t} vs oracle", t)
            # shared stem: with branches 1 and 2 fed the same latent / image latents (as the edit loop does), the stem up to
            # the first cross-attention may run on [source, shared]; same result as the full three-branch stem
            smp = inp16["sample"].clone()
            smp[2] = smp[1]
            il = inp16["image_latents"].clone()
            il[2] = il[1]
            kw_s = dict(kw_n, image_latents=il.to(DEV))
            for t in (981, 1):
                pnp_utils.register_time(pipe, t)
                outs = []
                for shared in (False, True):
                    ctx = native._prepare_clip(3, Fr, hw_[0], hw_[1], kw_s["encoder_hidden_states"], kw_s["fps"], kw_s["image_latents"],
                                               kw_s["image_embeddings"])
                    ctx.t_buf.fill_(float(t))
                    ctx.shared_stem = shared
                    outs.append(native._forward_core(ctx, smp.to(DEV).contiguous()).clone())
                out.append(_res(f"unet {cfg_name} shared stem == full stem (t={t}, inject={t == 981})", outs[1][:, :4], outs[0][:, :4].float(),
                                0.0 if DEV != "cpu" else 2e-2))
        finally:
            _unhook_all(m, ("ocpu",) + (("o16",) if calibrate else ()), pipe)
    return out


def check_engine_reuse_across_clips():
    """Multi-clip jobs keep the step engines (static buffers + HIP graphs) and re-point them at the next clip
    (``pipeline._engine`` / ``_StepEngine.rebind``): clip B through engines captured for clip A must be BIT-equal to clip B on a
    fresh pipeline -- any step-invariant tensor left over from clip A would show here.  Covers inversion, the PnP edit with its
    [negative, editing]-only engine (schedules ending early) and the plain CFG loop."""
    from anyv2v_amd import pnp_utils
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    out = []
    native, _, ocfg = build_pair("mini", 1234)
    Fr, hw, n_steps = 4, 8, 6
    h = lambda x: x.half().to(DEV)

    def clip(seed):
        torch.manual_seed(seed)
        inp = config1_inputs(ocfg, 3, Fr, hw)
        return h(inp["sample"][:1]) * (1.0 + 0.1 * seed), h(inp["encoder_hidden_states"]) + 0.05 * seed, h(inp["image_embeddings"]) * (1 + 0.2 * seed), \
            h(inp["image_latents"]) - 0.1 * seed

    def run(pipe, c):
        lat0, ehs, ie, il = c
        pnp_utils.clear_time(pipe)
        pipe.register_modules(scheduler=DDIMInverseScheduler())
        traj = pipe.invert(prompt_embeds=ehs[:1], image_embeddings=ie[:1], image_latents=il[:1], height=hw * 8, width=hw * 8,
                           num_frames=Fr, num_inference_steps=n_steps, guidance_scale=1.0, target_fps=8, latents=lat0,
                           return_trajectory=True)
        T = max(traj.keys())
        sched = DDIMScheduler()
        sched.set_timesteps(n_steps)
        k = lambda r: sched.timesteps[: int(n_steps * r)]
        pnp_utils.register_conv_injection(pipe, k(0.2))
        pnp_utils.register_spatial_attention_pnp(pipe, k(0.4))
        pnp_utils.register_temp_attention_pnp(pipe, k(0.5))
        pipe.register_modules(scheduler=sched)
        ed = pipe.sample_with_pnp(prompt_embeds=ehs[2:3], negative_prompt_embeds=ehs[1:2], image_embeddings=ie[2:3],
                                  image_latents=il[2:3], height=hw * 8, width=hw * 8, num_frames=Fr, num_inference_steps=n_steps,
                                  guidance_scale=9.0, target_fps=8, latents=traj[T].clone(), output_type="latent",
                                  ddim_init_latents_t_idx=0, ddim_inv_latents_path=traj, ddim_inv_prompt_embeds=ehs[:1],
                                  ddim_inv_image_embeddings=ie[:1], ddim_inv_image_latents=il[:1]).frames
        pnp_utils.clear_time(pipe)
        rec = pipe(prompt_embeds=ehs[:1], negative_prompt_embeds=ehs[1:2], image_embeddings=ie[:1], image_latents=il[:1],
                   height=hw * 8, width=hw * 8, num_frames=Fr, num_inference_steps=n_steps, guidance_scale=9.0, target_fps=8,
                   latents=traj[T].clone(), output_type="latent", ddim_init_latents_t_idx=0).frames
        return traj[T].clone(), ed.clone(), rec.clone()

    saved = os.environ.get("ANYV2V_ENGINE_CACHE")
    try:
        os.environ["ANYV2V_ENGINE_CACHE"] = "1"
        pipe = I2VGenXLPipeline(unet=native, scheduler=DDIMInverseScheduler())
        pipe._device = torch.device(DEV)
        a1 = run(pipe, clip(1))
        engines = {k: id(v) for k, v in pipe._engines.items()}
        b1 = run(pipe, clip(2))
        reused = {k: id(v) for k, v in pipe._engines.items()} == engines and len(engines) == 3
        a2 = run(pipe, clip(1))
        os.environ["ANYV2V_ENGINE_CACHE"] = "0"
        fresh = I2VGenXLPipeline(unet=native, scheduler=DDIMInverseScheduler())
        fresh._device = torch.device(DEV)
        b0 = run(fresh, clip(2))
        a0 = run(fresh, clip(1))
    finally:
        if saved is None:
            os.environ.pop("ANYV2V_ENGINE_CACHE", None)
        else:
            os.environ["ANYV2V_ENGINE_CACHE"] = saved
    out.append(dict(name="engines of clip A are the ones clip B runs on (3 kept: inversion, PnP edit, CFG)", err=0.0 if reused else 1.0,
                    l2=0.0, tol=0.0, ok=bool(reused)))
    for tag, got, ref in (("clip B on clip A's engines", b1, b0), ("clip A again after clip B", a2, a0), ("clip A, first use", a1, a0)):
        for name, g, r in zip(("inversion", "PnP edit", "CFG reconstruction"), got, ref):
            out.append(_res(f"engine reuse: {tag}: {name} == fresh pipeline (bitwise)", g, r.float(), 0.0))
    # weights reloaded between clips: the kept engines (their graphs point at the old packed tensors) must not be reused
    try:
        os.environ["ANYV2V_ENGINE_CACHE"] = "1"
        sd = {k: v.clone() for k, v in native.state_dict().items()}
        sd2 = {k: (v * 1.01 if v.is_floating_point() and v.dim() >= 2 else v) for k, v in sd.items()}
        native.load_state_dict(sd2)
        c1 = run(pipe, clip(1))
        os.environ["ANYV2V_ENGINE_CACHE"] = "0"
        c0 = run(fresh, clip(1))
        native.load_state_dict(sd)
    finally:
        if saved is None:
            os.environ.pop("ANYV2V_ENGINE_CACHE", None)
        else:
            os.environ["ANYV2V_ENGINE_CACHE"] = saved
    for name, g, r in zip(("inversion", "PnP edit", "CFG reconstruction"), c1, c0):
        out.append(_res(f"engine reuse: after load_state_dict the kept engines are rebuilt: {name} == fresh pipeline", g, r.float(), 0.0))
    changed = float((c0[0].float() - a0[0].float()).abs().max()) > 1e-4
    out.append(dict(name="engine reuse: the reloaded weights do change the result", err=0.0 if changed else 1.0, l2=0.0, tol=0.0, ok=bool(changed)))
    differs = float((b0[1].float() - a0[1].float()).abs().max()) > 1e-3
    out.append(dict(name="engine reuse: the two clips do differ", err=0.0 if differs else 1.0, l2=0.0, tol=0.0, ok=bool(differs)))
    return out



def check_loops_mini():
    """Multi-step: inversion -> PnP edit with the pipeline (HIP graphs) vs the oracle loops; plus graph == eager."""
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    from anyv2v_amd import pnp_utils
    from oracle import pnp_oracle
    out = []
    native, oracle, ocfg = build_pair("mini", 1234)
    Fr, hw, n_steps = 4, 8, 10
    inp = config1_inputs(ocfg, 3, Fr, hw)
    h = lambda x: x.half()
    lat0 = h(inp["sample"][:1])
    ehs, ie, il = h(inp["encoder_hidden_states"]), h(inp["image_embeddings"]), h(inp["image_latents"])
    # oracle: inversion with the source conditioning (branch 0), then PnP edit
    cond_src = dict(fps=torch.tensor([8]), image_latents=il[:1].float(), image_embeddings=ie[:1].float(),
                    encoder_hidden_states=ehs[:1].float())
    traj_o = pnp_oracle.invert_loop(oracle, lat0.float(), cond_src, n_steps)
    ie_all = torch.cat([ie[:1], torch.zeros_like(ie[2:3]), ie[2:3]])
    il_all = torch.cat([il[:1], il[2:3], il[2:3]])
    cond_all = dict(fps=torch.tensor([8] * 3), image_latents=il_all.float(), image_embeddings=ie_all.float(),
                    encoder_hidden_states=ehs.float())
    pnp_oracle.init_pnp(oracle, n_steps, 0.3, 0.6, 1.0)
    T = max(traj_o.keys())
    edited_o = pnp_oracle.pnp_loop(oracle, traj_o[T].clone(), traj_o, cond_all, n_steps, 9.0, t_idx=0)
    # calibration (GPU only): the same loops with the torch-eager fp16 oracle on this GPU; every multi-step bound below is
    # 2 x its drift against the fp32 oracle
    cal = DEV != "cpu"
    if cal:
        o16 = full_models("mini", 1234, want=("o16",))["o16"]
        d16 = lambda c: {k: (v.to(DEV, torch.float16) if v.is_floating_point() else v.to(DEV)) for k, v in c.items()}
        pnp_oracle.clear_hooks(o16)
        traj_e = pnp_oracle.invert_loop(o16, lat0.to(DEV), d16(cond_src), n_steps)
        pnp_oracle.init_pnp(o16, n_steps, 0.3, 0.6, 1.0)
        edited_e = pnp_oracle.pnp_loop(o16, traj_e[T].clone(), traj_e, d16(cond_all), n_steps, 9.0, t_idx=0)
        pnp_oracle.init_pnp(o16, n_steps, 0.2, 0.4, 0.5)
        edited_e2 = pnp_oracle.pnp_loop(o16, traj_e[T].clone(), traj_e, d16(cond_all), n_steps, 9.0, t_idx=0)
        pnp_oracle.clear_hooks(o16)
    bound = lambda name, got, ref, eager, tol: _calibrated(name, got, ref, eager, floor=1e-3) if cal else _res(name, got.cpu(), ref, tol)
    # native pipeline
    for graphs in ((True, False) if DEV != "cpu" else (False,)):
        os.environ["ANYV2V_NO_GRAPH"] = "0" if graphs else "1"
        pipe = I2VGenXLPipeline(unet=native, scheduler=DDIMInverseScheduler())
        pipe._device = torch.device(DEV)
        traj = pipe.invert(prompt_embeds=ehs[:1].to(DEV), image_embeddings=ie[:1].to(DEV), image_latents=il[:1].to(DEV),
                           height=hw * 8, width=hw * 8, num_frames=Fr, num_inference_steps=n_steps, guidance_scale=1.0,
                           target_fps=8, latents=lat0.to(DEV), return_trajectory=True)
        tag = "graph" if graphs else "eager"
        out.append(bound(f"pipeline.invert {n_steps} steps [{tag}] final latent vs oracle", traj[T], traj_o[T], traj_e[T] if cal else None, 5e-2))
        sched = DDIMScheduler()
        sched.set_timesteps(n_steps)
        k = lambda r: sched.timesteps[: int(n_steps * r)]
        pnp_utils.register_conv_injection(pipe, k(0.3))
        pnp_utils.register_spatial_attention_pnp(pipe, k(0.6))
        pnp_utils.register_temp_attention_pnp(pipe, k(1.0))
        pipe.register_modules(scheduler=sched)
        res = pipe.sample_with_pnp(prompt_embeds=ehs[2:3].to(DEV), negative_prompt_embeds=ehs[1:2].to(DEV),
                                   image_embeddings=ie[2:3].to(DEV), image_latents=il[2:3].to(DEV), height=hw * 8,
                                   width=hw * 8, num_frames=Fr, num_inference_steps=n_steps, guidance_scale=9.0,
                                   target_fps=8, latents=traj[T].clone(), output_type="latent", ddim_init_latents_t_idx=0,
                                   ddim_inv_latents_path=traj, ddim_inv_prompt_embeds=ehs[:1].to(DEV),
                                   ddim_inv_image_embeddings=ie[:1].to(DEV), ddim_inv_image_latents=il[:1].to(DEV)).frames
        out.append(bound(f"pipeline.sample_with_pnp {n_steps} steps [{tag}] vs oracle", res, edited_o, edited_e if cal else None, 8e-2))
        if graphs:
            res_graph = res.clone()
        elif DEV != "cpu":
            out.append(_res("pipeline graph replay == eager", res_graph.cpu(), res.cpu(), 1e-3))  # no atomics anywhere: deterministic
        if graphs or DEV == "cpu":
            # schedules that end early: the last steps lie outside every schedule, where the pipeline runs the
            # [negative, editing] branches only (exact: nothing reads the source branch there)
            def edit(skip):
                os.environ["ANYV2V_SRC_SKIP"] = "1" if skip else "0"
                pnp_utils.register_conv_injection(pipe, k(0.2))
                pnp_utils.register_spatial_attention_pnp(pipe, k(0.4))
                pnp_utils.register_temp_attention_pnp(pipe, k(0.5))
                return pipe.sample_with_pnp(prompt_embeds=ehs[2:3].to(DEV), negative_prompt_embeds=ehs[1:2].to(DEV),
                                            image_embeddings=ie[2:3].to(DEV), image_latents=il[2:3].to(DEV), height=hw * 8,
                                            width=hw * 8, num_frames=Fr, num_inference_steps=n_steps, guidance_scale=9.0,
                                            target_fps=8, latents=traj[T].clone(), output_type="latent",
                                            ddim_init_latents_t_idx=0, ddim_inv_latents_path=traj,
                                            ddim_inv_prompt_embeds=ehs[:1].to(DEV), ddim_inv_image_embeddings=ie[:1].to(DEV),
                                            ddim_inv_image_latents=il[:1].to(DEV)).frames
            r_skip, r_full = edit(True), edit(False)
            os.environ["ANYV2V_SRC_SKIP"] = "1"
            # the source branch's dead tail (behind the last hook site) dropped / computed: same latents.  On the GPU bit for bit
            # (the shortened launches run under the batch hint (3, 2)); the CPU emulation's BLAS depends on M
            os.environ["ANYV2V_DROP_SRC_TAIL"] = "0"
            r_tail = edit(True)
            os.environ["ANYV2V_DROP_SRC_TAIL"] = "1"
            if DEV == "cpu":
                out.append(_res(f"source-branch tail dropped == computed [{tag}]", r_skip.cpu(), r_tail.cpu(), 2e-2))
            else:
                out.append(dict(name=f"source-branch tail dropped == computed, bit for bit [{tag}]", tol=0.0,
                                err=float((r_skip.float() - r_tail.float()).abs().max()), ok=bool(torch.equal(r_skip, r_tail))))
            pnp_oracle.init_pnp(oracle, n_steps, 0.2, 0.4, 0.5)
            edited_o2 = pnp_oracle.pnp_loop(oracle, traj_o[T].clone(), traj_o, cond_all, n_steps, 9.0, t_idx=0)
            out.append(bound(f"sample_with_pnp early-ending schedules [{tag}] vs oracle", r_skip, edited_o2, edited_e2 if cal else None, 8e-2))
            out.append(_res(f"source-branch skip on schedule-free steps == full B=3 steps [{tag}]", r_skip.cpu(), r_full.cpu(), 2e-2))  # not bitwise: GEMM tiling (and the CPU BLAS in the emulation) depends on M
    os.environ["ANYV2V_NO_GRAPH"] = "0"
    # A3: plain CFG sampling (DDIM reconstruction), B = 2 [negative, positive]; shared stem on / off
    cond2 = dict(fps=torch.tensor([8] * 2), image_latents=torch.cat([il[2:3], il[2:3]]).float(),
                 image_embeddings=torch.cat([torch.zeros_like(ie[2:3]), ie[2:3]]).float(),
                 encoder_hidden_states=torch.cat([ehs[1:2], ehs[2:3]]).float())
    _, oracle_plain, _ = build_pair("mini", 1234)  # same weights, no hooks registered
    rec_o = pnp_oracle.sample_loop(oracle_plain, traj_o[T].clone(), cond2, n_steps, 9.0, t_idx=0)
    if cal:
        rec_e = pnp_oracle.sample_loop(o16, traj_o[T].clone().to(DEV, torch.float16), d16(cond2), n_steps, 9.0, t_idx=0)
    recs = []
    for shared in ("1", "0"):
        os.environ["ANYV2V_SHARED_STEM"] = shared
        pipe = I2VGenXLPipeline(unet=native, scheduler=DDIMScheduler())
        pipe._device = torch.device(DEV)
        recs.append(pipe(prompt_embeds=ehs[2:3].to(DEV), negative_prompt_embeds=ehs[1:2].to(DEV), image_embeddings=ie[2:3].to(DEV),
                         image_latents=il[2:3].to(DEV), height=hw * 8, width=hw * 8, num_frames=Fr, num_inference_steps=n_steps,
                         guidance_scale=9.0, target_fps=8, latents=traj_o[T].clone().half().to(DEV), output_type="latent",
                         ddim_init_latents_t_idx=0).frames)
    os.environ["ANYV2V_SHARED_STEM"] = "1"
    out.append(bound(f"pipeline.__call__ (CFG sampling) {n_steps} steps vs oracle", recs[0], rec_o, rec_e if cal else None, 8e-2))
    out.append(_res("CFG sampling: shared stem == separate stems", recs[0].cpu(), recs[1].cpu().float(), 0.0 if DEV != "cpu" else 2e-2))
    return out


# ------------------------------------------------------------------------------------------------ N1: full model, benchmarked sizes
_MODELS = {}


def check_source_cache(cfg_name="mini", Fr=4, hw=8, n_steps=6):
    """Multi-edit job (``pipeline.SourceFeatureCache``): three edits of ONE clip -- different prompts and edited frames, the third
    also with shorter injection schedules -- with the cache (the first records, the others replay and run [negative, editing]
    only) must be BIT-EQUAL to the same three edits without it; the cache really is used (replayed steps counted) and is dropped
    when the clip changes."""
    from anyv2v_amd import pnp_utils
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    out = []
    if cfg_name == "full":   # (the full-width model without its CPU oracle: this check compares the product with itself)
        m = full_models("full", 1234, want=("native",))
        native, ocfg = m["native"], m["ocfg"]
    else:
        native, _, ocfg = build_pair(cfg_name, 1234)
    inp = config1_inputs(ocfg, 3, Fr, hw)
    g = lambda x: x.half().to(DEV)
    lat0, ehs, ie, il = g(inp["sample"][:1]), g(inp["encoder_hidden_states"]), g(inp["image_embeddings"]), g(inp["image_latents"])
    gen = torch.Generator().manual_seed(77)
    edits = []
    for k in range(3):   # (prompt embeds, negative embeds, image embedding, image latents of the edited frame, schedule ratios)
        r = lambda *sh: torch.randn(*sh, generator=gen).half().to(DEV)
        il_k = il[2:3].clone()
        il_k[:, :, 0] = r(*il_k[:, :, 0].shape)
        edits.append((r(*ehs[2:3].shape), ehs[1:2], r(*ie[2:3].shape), il_k, (0.5, 0.67, 1.0) if k < 2 else (0.2, 0.34, 0.5)))

    def run_all(use_cache):
        pipe = I2VGenXLPipeline(unet=native, scheduler=DDIMInverseScheduler())
        pipe._device = torch.device(DEV)
        cache = pipe.enable_source_cache(True) if use_cache else None
        traj = pipe.invert(prompt_embeds=ehs[:1], image_embeddings=ie[:1], image_latents=il[:1], height=hw * 8, width=hw * 8,
                           num_frames=Fr, num_inference_steps=n_steps, guidance_scale=1.0, target_fps=8, latents=lat0,
                           return_trajectory=True)
        T = max(traj.keys())
        res = []
        for (pe, npe, ie_k, il_k, ratios) in edits:
            sched = DDIMScheduler()
            sched.set_timesteps(n_steps)
            pipe.register_modules(scheduler=sched)
            k = lambda r_: sched.timesteps[: int(n_steps * r_)]
            pnp_utils.register_conv_injection(pipe, k(ratios[0]))
            pnp_utils.register_spatial_attention_pnp(pipe, k(ratios[1]))
            pnp_utils.register_temp_attention_pnp(pipe, k(ratios[2]))
            res.append(pipe.sample_with_pnp(prompt_embeds=pe, negative_prompt_embeds=npe, image_embeddings=ie_k, image_latents=il_k,
                                            height=hw * 8, width=hw * 8, num_frames=Fr, num_inference_steps=n_steps,
                                            guidance_scale=9.0, target_fps=8, latents=traj[T].clone(), output_type="latent",
                                            ddim_init_latents_t_idx=0, ddim_inv_latents_path=traj, ddim_inv_prompt_embeds=ehs[:1],
                                            ddim_inv_image_embeddings=ie[:1], ddim_inv_image_latents=il[:1]).frames.float().cpu())
            pnp_utils.clear_time(pipe)
        return res, cache, pipe, traj

    plain, _, _, _ = run_all(False)
    cached, cache, pipe, traj = run_all(True)
    # On the GPU the kernels' arithmetic per output element does not depend on the batch, so the replayed two-branch steps are
    # BIT-equal to the three-branch ones.  The CPU op emulation (torch matmul picks its summation order by shape) is not: there
    # the first edit (recorded, three-branch) must still be bit-equal and the replayed ones agree to fp16 rounding amplified by
    # cfg 9 over the steps -- a wrong feature at any site is an O(1) error (the single-forward check below is the sharp one).
    exact = DEV != "cpu"
    for k in range(3):
        out.append(_res(f"source cache: edit {k} with the cache == without ({'bit-equal' if exact or k == 0 else 'fp16 drift'})",
                        cached[k], plain[k], 0.0 if (exact or k == 0) else 0.12))
    # one forward, every site type on: [negative, editing] with replayed source features vs branches 1, 2 of the three-branch forward
    import types
    smp = torch.cat([lat0, lat0 * 0.7, lat0 * 0.7])
    il3 = torch.cat([il[:1], il[2:3], il[2:3]])
    kw = dict(fps=torch.tensor([8, 8, 8], device=DEV), image_latents=il3, image_embeddings=ie, encoder_hidden_states=ehs)
    p2 = types.SimpleNamespace(unet=native)
    tsl = [981 - 20 * i for i in range(50)]
    pnp_utils.register_conv_injection(p2, tsl)
    pnp_utils.register_spatial_attention_pnp(p2, tsl)
    pnp_utils.register_temp_attention_pnp(p2, tsl)
    pnp_utils.register_time(p2, 981)
    sites = pnp_utils.injection_sites(p2)
    full = Fr * hw * hw
    bufs = {n: torch.zeros((full // {"1": 16, "2": 4, "3": 1}[n.split(".up")[1][0]], c), dtype=torch.float16, device=DEV) for n, _o, c in sites}
    for n, o, _c in sites:
        o.src_io = ("record", bufs[n])
    v3 = native(smp, 981, **kw)[0].float().cpu()
    for n, o, _c in sites:
        o.src_io = ("replay", bufs[n])
    with ops.batch_hint(3, 2):   # (what the replay engine runs under: every launch plans as the three-branch launch it stands for)
        v2 = native(smp[1:], 981, **{k_: v_[1:] for k_, v_ in kw.items()})[0].float().cpu()
    for n, o, _c in sites:
        o.src_io = None
    pnp_utils.clear_time(p2)
    out.append(_res("source cache: one forward, 17 sites replayed ([negative, editing]) vs the three-branch forward", v2, v3[1:],
                    0.0 if exact else 4e-3))
    out.append(_res("source cache: the edits differ from each other (not vacuous)", (plain[0] - plain[1]).abs().max().reshape(1),
                    torch.zeros(1), float("inf")))
    out[-1]["ok"] = bool((plain[0] - plain[1]).abs().max() > 1e-3)
    n_inj = n_steps  # schedules of edit 0 / 1 cover every step at ratio 1.0 for the temporal sites
    ok = cache.recorded_steps >= n_inj and cache.replayed_steps >= n_inj
    out.append(dict(name=f"source cache: recorded {cache.recorded_steps} steps, replayed {cache.replayed_steps} ({cache.nbytes() / 2**20:.1f} MiB)",
                    err=0.0 if ok else 1.0, l2=0.0, tol=0.5, ok=ok))
    # another clip (other trajectory object): the signature changes, nothing of the old clip is replayed
    before = cache.replayed_steps
    traj2 = pipe.invert(prompt_embeds=ehs[:1], image_embeddings=ie[:1], image_latents=il[:1], height=hw * 8, width=hw * 8,
                        num_frames=Fr, num_inference_steps=n_steps, guidance_scale=1.0, target_fps=8, latents=lat0 * 0.5,
                        return_trajectory=True)
    sched = DDIMScheduler()
    sched.set_timesteps(n_steps)
    pipe.register_modules(scheduler=sched)
    pnp_utils.register_conv_injection(pipe, sched.timesteps)
    pnp_utils.register_spatial_attention_pnp(pipe, sched.timesteps)
    pnp_utils.register_temp_attention_pnp(pipe, sched.timesteps)
    pe, npe, ie_k, il_k, _ = edits[0]
    pipe.sample_with_pnp(prompt_embeds=pe, negative_prompt_embeds=npe, image_embeddings=ie_k, image_latents=il_k, height=hw * 8,
                         width=hw * 8, num_frames=Fr, num_inference_steps=n_steps, guidance_scale=9.0, target_fps=8,
                         latents=traj2[max(traj2.keys())].clone(), output_type="latent", ddim_init_latents_t_idx=0,
                         ddim_inv_latents_path=traj2, ddim_inv_prompt_embeds=ehs[:1], ddim_inv_image_embeddings=ie[:1],
                         ddim_inv_image_latents=il[:1])
    pnp_utils.clear_time(pipe)
    ok = cache.replayed_steps == before
    out.append(dict(name="source cache: a new clip empties it (no replay of the old clip's features)", err=0.0 if ok else 1.0, l2=0.0,
                    tol=0.5, ok=ok))
    return out


def full_models(cfg_name="full", seed=1234, want=("native", "o32", "o16")):
    """One set of models per process, shared by the full-size parity checks: the native HIP UNet, the oracle in fp32 on
    the GPU (the CHECKER at sizes the CPU cannot reach in test time; validated against the CPU oracle in
    ``check_n1_config1``), the same oracle run by PyTorch-ROCm eager in fp16 (the stand-in for "the reference's fp16
    path", SURVEY.md 8(c): calibrates every tolerance), and the fp32 CPU oracle -- all from ONE seeded fp16-rounded
    state dict."""
    from anyv2v_amd.unet import I2VGenXLUNet, I2VGenXLUNetConfig
    from oracle.unet_oracle import UNetConfig, build_oracle, random_state_dict
    key = (cfg_name, seed)
    m = _MODELS.setdefault(key, {})
    ocfg = UNetConfig.mini() if cfg_name == "mini" else UNetConfig.i2vgen_xl()
    m["ocfg"] = ocfg
    if any(w not in m for w in want) and "sd" not in m:
        m["sd"] = random_state_dict(ocfg, seed)
    for w in want:
        if w in m:
            continue
        if w == "native":
            ncfg = I2VGenXLUNetConfig.mini() if cfg_name == "mini" else I2VGenXLUNetConfig()
            with torch.device("meta"):
                n = I2VGenXLUNet(ncfg)
            n = n.to_empty(device=DEV)
            n.load_state_dict({k: v.to(DEV) for k, v in m["sd"].items()}, strict=True)
            m[w] = n
        elif w == "o32":
            m[w] = build_oracle(ocfg, m["sd"], dtype=torch.float32, device=DEV)
        elif w == "o16":
            m[w] = build_oracle(ocfg, m["sd"], dtype=torch.float16, device=DEV)
        elif w == "ocpu":
            m[w] = build_oracle(ocfg, m["sd"], dtype=torch.float32, device="cpu")
    return m


def release_models():
    _MODELS.clear()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def _sync():
    if DEV != "cpu":
        torch.cuda.synchronize()


_CAPS = None


def _recorded_caps():
    """tests/golden/hip_error_caps.json: the HIP path's own error per whole-model check as last recorded on an MI355X
    (``ANYV2V_RECORD_ERRS=<file> pytest -m gpu`` / ``tools/gpu_check.py`` writes it).  A calibrated bound is never looser than
    3 x that record, so a 3 x regression of the HIP path fails even where eager fp16 is far worse."""
    global _CAPS
    if _CAPS is None:
        import json
        path = os.path.join(ROOT, "tests", "golden", "hip_error_caps.json")
        _CAPS = json.load(open(path)) if os.path.isfile(path) else {}
    return _CAPS


_RECORD = {}


def _calibrated(name, got, ref, eager, factor=2.0, floor=5e-4, key=None, gap_cap=None, max_tol=None):
    """SURVEY.md 8(c) tolerance policy: |HIP fp16 - fp32 oracle| <= 2 x |torch-eager fp16 oracle - fp32 oracle| (+ a floor
    for the cases where both are at rounding level), all three on the same inputs in the same test -- for the max-abs metric AND
    for relative L2 -- and, with ``key``, additionally <= 3 x the HIP error recorded for this check (``_recorded_caps``).
    Every row also reports |HIP - eager fp16| (same normalisation): when both fp16 paths sit far from the fp32 checker but close to
    each other, the CHECKER is the outlier; ``gap_cap`` bounds that distance.  ``max_tol``: a fixed bound the
    calibrated one may not exceed (a calibration arm that shares the model code must not LOOSEN an older fixed bound)."""
    got, ref, eager = got.float().cpu(), ref.float().cpu(), eager.float().cpu()
    if not torch.isfinite(got).all():
        return dict(name=name, err=float("nan"), l2=float("nan"), tol=0.0, ok=False)
    e_h, l2 = _rel(got, ref)
    e_e, l2_e = _rel(eager, ref)
    e_g, l2_g = _rel(got, eager)
    tol = factor * e_e + floor
    tol_l2 = factor * l2_e + floor
    if max_tol is not None:
        tol, tol_l2 = min(tol, max_tol), min(tol_l2, max_tol)
    cap = None
    if key is not None:
        _RECORD[key] = {"err": e_h, "l2": l2, "eager": e_e, "eager_l2": l2_e, "hip_vs_eager": e_g, "hip_vs_eager_l2": l2_g}
        rec = _recorded_caps().get(key)
        if rec is not None:
            cap = 3.0 * rec["err"] + floor
            tol, tol_l2 = min(tol, cap), min(tol_l2, 3.0 * rec["l2"] + floor)
        path = os.environ.get("ANYV2V_RECORD_ERRS")
        if path:
            import json
            json.dump(_RECORD, open(path, "w"), indent=1, sort_keys=True)
    ok = bool(e_h <= tol and l2 <= tol_l2 and (gap_cap is None or e_g <= gap_cap))
    return dict(name=f"{name}  [HIP {e_h:.2e} (l2 {l2:.2e}) | eager fp16 {e_e:.2e} (l2 {l2_e:.2e}) | HIP-vs-eager {e_g:.2e} (l2 {l2_g:.2e})"
                     + (f" <= {gap_cap:.2e}" if gap_cap is not None else "") + (f" | cap {cap:.2e}" if cap else "") + "]",
                err=e_h, l2=l2, tol=tol, tol_l2=tol_l2, ok=ok, eager=e_e, gap=e_g)


def _emulated(fn):
    """Runs ``fn`` with ``anyv2v_amd.ops`` replaced by tests/cpu_ops_emulation.py (plain torch on the CPU, fp32 arithmetic with the
    kernels' fp16 storage points, no HIP graphs) and restores the kernels: an INDEPENDENT fp16-storage implementation of the same model
    on the same weights -- the "eager fp16" arm of ``_calibrated`` for the sibling pipelines, whose reference classes do not exist on
    the GPU box (the fixtures hold their fp32 outputs)."""
    import cpu_ops_emulation as emu
    names = ["gemm", "groupnorm", "layernorm", "softmax_rows", "attention", "silu", "add", "timestep_embedding", "ncfhw_to_tokens",
             "tokens_to_ncfhw", "adaptive_avgpool", "copy_cols", "gather_rows", "rotary", "cfg_ddim_step", "ddim_step", "guided_step", "ff_geglu"]
    saved = {n: getattr(ops, n) for n in names}
    saved_env = os.environ.get("ANYV2V_NO_GRAPH")
    os.environ["ANYV2V_NO_GRAPH"] = "1"
    emu.install()
    try:
        return fn()
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
        if saved_env is None:
            os.environ.pop("ANYV2V_NO_GRAPH", None)
        else:
            os.environ["ANYV2V_NO_GRAPH"] = saved_env


def _cond_kw(inp16, device, dtype):
    return dict(fps=inp16["fps"].to(device), image_latents=inp16["image_latents"].to(device, dtype),
                image_embeddings=inp16["image_embeddings"].to(device, dtype),
                encoder_hidden_states=inp16["encoder_hidden_states"].to(device, dtype))


def _hook_all(models, names, n_steps=50, ratios=(0.2, 0.5, 0.8)):
    """Native hooks on the native UNet, ``oracle.pnp_oracle`` hooks on the oracles; same schedules (10 / 25 / 40 steps of 50:
    t=981 on all 17 sites, t=701 spatial + temporal, t=301 temporal only, t=101 none)."""
    import types
    from anyv2v_amd import pnp_utils
    from oracle import pnp_oracle
    ts = [981 - 20 * i for i in range(50)] if n_steps == 50 else None
    assert ts is not None
    k = lambda r: ts[: int(n_steps * r)]
    pipe = types.SimpleNamespace(unet=models["native"])
    pnp_utils.register_conv_injection(pipe, k(ratios[0]))
    pnp_utils.register_spatial_attention_pnp(pipe, k(ratios[1]))
    pnp_utils.register_temp_attention_pnp(pipe, k(ratios[2]))
    for n in names:
        pnp_oracle.register_conv_injection(models[n], k(ratios[0]))
        pnp_oracle.register_spatial_attention_pnp(models[n], k(ratios[1]))
        pnp_oracle.register_temp_attention_pnp(models[n], k(ratios[2]))
    return pipe


def _unhook_all(models, names, pipe):
    from anyv2v_amd import pnp_utils
    from oracle import pnp_oracle
    pnp_utils.clear_time(pipe)
    for n in names:
        pnp_oracle.clear_hooks(models[n])


def check_n1_config1(cfg_name="full", Fr=8, hw=32, golden=True, report=None):
    """VERDICT r1 N1 (a)+(b): the FULL 1.42 B model at BASELINE config 1 (8 f x 256^2), B=1 and B=3 with all 17 hook sites
    registered, t in {981, 301} -- the exact ``cpu_baseline`` workload -- HIP fp16 vs (i) the fp32 CPU oracle driven by
    ``oracle.pnp_oracle``, (ii) the full-width fixture generated by the REFERENCE's own ``pnp_utils.py`` on the oracle
    (``tests/golden/make_golden.py`` -> ``pnp_hooks_full_config1.pt``); tolerance = 2 x the eager-fp16 error measured here.
    Also validates the GPU-resident fp32 oracle (the checker of the config-3 tests) against the CPU oracle."""
    import time
    from anyv2v_amd import pnp_utils
    from oracle import pnp_oracle
    out = []
    m = full_models(cfg_name, 1234, want=("native", "o32", "o16", "ocpu"))
    native, o32, o16, ocpu, ocfg = m["native"], m["o32"], m["o16"], m["ocpu"], m["ocfg"]
    inp = config1_inputs(ocfg, 3, Fr, hw)
    inp16 = {k: (v.half() if v.is_floating_point() else v) for k, v in inp.items()}
    sl = lambda d, s: {k: v[s] for k, v in d.items()}
    smp = inp16["sample"]

    def run_all(B, t):
        i16 = sl(inp16, slice(0, B))
        t0 = time.time()
        with torch.no_grad():
            v_cpu = ocpu(i16["sample"].float(), t, **_cond_kw(i16, "cpu", torch.float32))[0]
            t_cpu = time.time() - t0
            v32 = o32(i16["sample"].float().to(DEV), t, **_cond_kw(i16, DEV, torch.float32))[0]
            v16 = o16(i16["sample"].to(DEV), t, **_cond_kw(i16, DEV, torch.float16))[0]
        vn = native(i16["sample"].to(DEV), t, **_cond_kw(i16, DEV, torch.float16))[0]
        return v_cpu, v32, v16, vn, t_cpu

    v_cpu, v32, v16, vn, t_cpu = run_all(1, 981)
    if report is not None:
        report[f"cpu_oracle_seconds_{cfg_name}_B1"] = t_cpu
    out.append(_res(f"fp32 oracle on the GPU == fp32 oracle on the CPU ({cfg_name}, config 1, B=1)", v32.cpu(), v_cpu, 2e-4))
    out.append(_calibrated(f"unet {cfg_name} config 1 B=1 t=981 vs CPU fp32 oracle", vn, v_cpu, v16, key=f"n1c1:{cfg_name}:F{Fr}x{hw}:B1:t981"))
    pipe = _hook_all(m, ("o32", "o16", "ocpu"))
    gold = None
    gpath = os.path.join(ROOT, "tests", "golden", "pnp_hooks_full_config1.pt")
    if golden and cfg_name == "full" and (Fr, hw) == (8, 32):
        gold = torch.load(gpath)
        assert gold["weights_seed"] == 1234 and gold["input_seed"] == 8888 and tuple(gold["shape"]) == (3, 4, Fr, hw, hw)
    try:
        for t in (981, 301):
            pnp_utils.register_time(pipe, t)
            for n in ("o32", "o16", "ocpu"):
                pnp_oracle.register_time(m[n], t)
            v_cpu, v32, v16, vn, t_cpu = run_all(3, t)
            if report is not None:
                report[f"cpu_oracle_seconds_{cfg_name}_B3_t{t}"] = t_cpu
            out.append(_res(f"fp32 oracle GPU == CPU ({cfg_name}, config 1, B=3 + hooks, t={t})", v32.cpu(), v_cpu, 2e-4))
            out.append(_calibrated(f"unet {cfg_name} config 1 B=3 + 17 hook sites t={t} vs CPU fp32 oracle (pnp_oracle)", vn, v_cpu, v16,
                                   key=f"n1c1:{cfg_name}:F{Fr}x{hw}:B3:t{t}"))
            if gold is not None:
                out.append(_res(f"CPU oracle + pnp_oracle == fixture of the reference's pnp_utils on the oracle, t={t}", v_cpu,
                                gold[f"v_hook_t{t}"], 1e-4))
                out.append(_calibrated(f"unet full config 1 B=3 + hooks t={t} vs reference-pnp_utils fixture", vn,
                                       gold[f"v_hook_t{t}"], v16, key=f"n1c1:fixture:t{t}"))
    finally:
        _unhook_all(m, ("o32", "o16", "ocpu"), pipe)
    return out


def check_n1_config3_step(cfg_name="full", Fr=16, hw=64, report=None, hooked_ts=(981, 301), chunked=None, plain_fp32_too=False,
                          batches=(1, 3)):
    """ONE step at the benchmarked size -- BASELINE config 3, latents [B,4,16,64,64] -- B=1 (inversion
    step) and B=3 with all 17 hook sites (PnP step; t=981 every site on, t=301 temporal only), HIP fp16 vs the fp32 oracle
    run by torch-eager on the GPU (checker only), tolerance 2 x the eager-fp16 oracle's error at this size.

    ``chunked`` (default: for clips longer than 16 frames, i.e. the config-5 geometry): both eager oracles are evaluated through
    ``oracle/chunked.py`` -- pieces no larger than the config-3 rows' tensors, same arithmetic per output element -- because a plain
    eager activation at [3,4,128,64,64] reaches 4.03 G elements / 16 GB (the un-chunked fp32 checker was the
    suspected outlier of the 0.142 row).  ``plain_fp32_too`` also runs the un-chunked fp32 oracle and reports it against the
    chunked one (diagnostic: locates the checker's own error).  The B=3 rows additionally bound |HIP - eager fp16| by 3 x the
    eager-vs-fp32 error of the B=1 row: the two fp16 implementations must stay as close to each other as one of them is to fp32."""
    import time
    from anyv2v_amd import pnp_utils
    from oracle import chunked as ch
    from oracle import pnp_oracle
    out = []
    m = full_models(cfg_name, 1234, want=("native", "o32", "o16"))
    native, o32, o16, ocfg = m["native"], m["o32"], m["o16"], m["ocfg"]
    inp = config1_inputs(ocfg, 3, Fr, hw)
    inp16 = {k: (v.half() if v.is_floating_point() else v) for k, v in inp.items()}
    sl = lambda d, s: {k: v[s] for k, v in d.items()}
    if chunked is None:
        chunked = Fr > 16
    tag = " (chunked eager oracles)" if chunked else ""

    def run_oracle(o, dt, i16, t, B, use_chunks):
        if use_chunks:
            ch.enable_chunking(o, B, Fr, frame_chunk=16, row_chunk=max(1, 65536 // (hw * Fr)))
        try:
            with torch.no_grad():
                x = i16["sample"].to(DEV, dt)
                return o(x, t, **_cond_kw(i16, DEV, dt))[0].cpu()
        finally:
            if use_chunks:
                ch.disable_chunking(o)

    def run_all(B, t):
        i16 = sl(inp16, slice(0, B))
        _sync()
        t0 = time.time()
        v32 = run_oracle(o32, torch.float32, i16, t, B, chunked)
        _sync()
        t32 = time.time() - t0
        v16 = run_oracle(o16, torch.float16, i16, t, B, chunked)
        vn = native(i16["sample"].to(DEV), t, **_cond_kw(i16, DEV, torch.float16))[0]
        _sync()
        if report is not None:
            report[f"gpu_fp32_oracle_seconds_{cfg_name}_B{B}_F{Fr}_{hw}"] = t32
        print(f"     [B={B} t={t}: fp32 oracle{tag} {t32:.1f} s]", flush=True)
        return v32, v16.cpu(), vn.cpu()

    gap_cap = None
    if 1 in batches:
        v32, v16, vn = run_all(1, 981)
        r = _calibrated(f"unet {cfg_name} [1,4,{Fr},{hw},{hw}] t=981 vs fp32 oracle{tag}", vn, v32, v16, key=f"n1step:{cfg_name}:F{Fr}x{hw}:B1:t981")
        out.append(r)
        gap_cap = 3.0 * r["eager"] + 5e-4
    if 3 not in batches:
        return out
    if gap_cap is None:   # B=1 row skipped: its recorded eager-fp16 error
        rec = _recorded_caps().get(f"n1step:{cfg_name}:F{Fr}x{hw}:B1:t981")
        gap_cap = 3.0 * rec["eager"] + 5e-4 if rec is not None else None
    pipe = _hook_all(m, ("o32", "o16"))
    try:
        for t in hooked_ts:
            pnp_utils.register_time(pipe, t)
            for n in ("o32", "o16"):
                pnp_oracle.register_time(m[n], t)
            v32, v16, vn = run_all(3, t)
            # (the distance between the two fp16 implementations cannot be asked to be smaller than the eager one's own error on THIS
            #  row: at [3,4,128,64,64] eager fp16 has isolated outliers -- max-rel 2.8e-2 at rel-L2 6.7e-3 -- that the HIP path, 2.1e-3 /
            #  2.1e-3 from fp32, does not share; profiles/r04_gputest_n1_log.txt)
            cap_t = None if gap_cap is None else max(gap_cap, 3.0 * _rel(v16, v32)[0] + 5e-4)
            out.append(_calibrated(f"unet {cfg_name} [3,4,{Fr},{hw},{hw}] + 17 hook sites t={t} vs fp32 oracle{tag}", vn, v32, v16,
                                   key=f"n1step:{cfg_name}:F{Fr}x{hw}:B3:t{t}", gap_cap=cap_t))
            if plain_fp32_too and chunked:
                i16 = sl(inp16, slice(0, 3))
                t0 = time.time()
                p32 = run_oracle(o32, torch.float32, i16, t, 3, False)
                _sync()
                e, l2 = _rel(p32, v32)
                print(f"     [diagnostic, B=3 t={t}: UN-chunked fp32 eager oracle vs chunked fp32 oracle: max-rel {e:.3e}, rel-L2 {l2:.3e} "
                      f"({time.time() - t0:.1f} s); HIP vs un-chunked fp32: {_rel(vn, p32)[0]:.3e}]", flush=True)
                p16 = run_oracle(o16, torch.float16, i16, t, 3, False)
                e, l2 = _rel(p16, v16)
                print(f"     [diagnostic, B=3 t={t}: UN-chunked fp16 eager oracle vs chunked fp16 eager oracle: max-rel {e:.3e}, rel-L2 {l2:.3e}]",
                      flush=True)
                if report is not None:
                    report[f"plain_fp32_vs_chunked_fp32_B3_F{Fr}_{hw}_t{t}"] = _rel(p32, v32)
                    report[f"plain_fp16_vs_chunked_fp16_B3_F{Fr}_{hw}_t{t}"] = (e, l2)
    finally:
        _unhook_all(m, ("o32", "o16"), pipe)
    return out


def check_n1_drift(cfg_name="full", Fr=16, hw=64, n_steps=50, every=10, report=None):
    """n-step DDIM inversion -> n-step PnP edit, and inversion -> plain CFG reconstruction, through the
    product pipeline (HIP graphs) vs the oracle loops in fp32 on the GPU; latent drift reported every ``every`` steps and
    bounded by 2 x the drift of the torch-eager fp16 oracle run through the SAME oracle loops.  Schedules 0.2 / 0.5 / 0.8
    (so injected, partly injected and injection-free steps all occur), cfg 9.0, seed-8888 inputs."""
    from anyv2v_amd import pnp_utils
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    from oracle import pnp_oracle
    out = []
    m = full_models(cfg_name, 1234, want=("native", "o32", "o16"))
    native, ocfg = m["native"], m["ocfg"]
    inp = config1_inputs(ocfg, 3, Fr, hw)
    h = lambda x: x.half()
    lat0 = h(inp["sample"][:1])
    ehs, ie, il = h(inp["encoder_hidden_states"]), h(inp["image_embeddings"]), h(inp["image_latents"])
    ie_all = torch.cat([ie[:1], torch.zeros_like(ie[2:3]), ie[2:3]])
    il_all = torch.cat([il[:1], il[2:3], il[2:3]])
    ratios = (0.2, 0.5, 0.8)

    def oracle_run(o, dt):
        d = lambda x: x.to(DEV, dt)
        cond_src = dict(fps=torch.tensor([8], device=DEV), image_latents=d(il[:1]), image_embeddings=d(ie[:1]),
                        encoder_hidden_states=d(ehs[:1]))
        cond_all = dict(fps=torch.tensor([8] * 3, device=DEV), image_latents=d(il_all), image_embeddings=d(ie_all),
                        encoder_hidden_states=d(ehs))
        cond2 = dict(fps=torch.tensor([8] * 2, device=DEV), image_latents=d(il_all[1:]), image_embeddings=d(ie_all[1:]),
                     encoder_hidden_states=d(ehs[1:]))
        pnp_oracle.clear_hooks(o)
        traj = pnp_oracle.invert_loop(o, d(lat0), cond_src, n_steps)
        T = max(traj.keys())
        rec = {}
        pnp_oracle.sample_loop(o, traj[T].clone(), cond2, n_steps, 9.0, t_idx=0, trace=rec)
        pnp_oracle.init_pnp(o, n_steps, *ratios)
        ed = {}
        pnp_oracle.pnp_loop(o, traj[T].clone(), traj, cond_all, n_steps, 9.0, t_idx=0, trace=ed)
        pnp_oracle.clear_hooks(o)
        return traj, rec, ed

    traj32, rec32, ed32 = oracle_run(m["o32"], torch.float32)
    traj16, rec16, ed16 = oracle_run(m["o16"], torch.float16)
    # product pipeline
    g = lambda x: x.to(DEV)
    pipe = I2VGenXLPipeline(unet=native, scheduler=DDIMInverseScheduler())
    pipe._device = torch.device(DEV)
    traj = pipe.invert(prompt_embeds=g(ehs[:1]), image_embeddings=g(ie[:1]), image_latents=g(il[:1]), height=hw * 8, width=hw * 8,
                       num_frames=Fr, num_inference_steps=n_steps, guidance_scale=1.0, target_fps=8, latents=g(lat0),
                       return_trajectory=True)
    T = max(traj.keys())
    sched = DDIMScheduler()
    sched.set_timesteps(n_steps)
    pipe.register_modules(scheduler=sched)
    rec = {}
    pipe(prompt_embeds=g(ehs[2:3]), negative_prompt_embeds=g(ehs[1:2]), image_embeddings=g(ie[2:3]), image_latents=g(il[2:3]),
         height=hw * 8, width=hw * 8, num_frames=Fr, num_inference_steps=n_steps, guidance_scale=9.0, target_fps=8,
         latents=traj[T].clone(), output_type="latent", ddim_init_latents_t_idx=0, latents_trace=rec)
    k = lambda r: sched.timesteps[: int(n_steps * r)]
    pnp_utils.register_conv_injection(pipe, k(ratios[0]))
    pnp_utils.register_spatial_attention_pnp(pipe, k(ratios[1]))
    pnp_utils.register_temp_attention_pnp(pipe, k(ratios[2]))
    ed = {}
    pipe.sample_with_pnp(prompt_embeds=g(ehs[2:3]), negative_prompt_embeds=g(ehs[1:2]), image_embeddings=g(ie[2:3]),
                         image_latents=g(il[2:3]), height=hw * 8, width=hw * 8, num_frames=Fr, num_inference_steps=n_steps,
                         guidance_scale=9.0, target_fps=8, latents=traj[T].clone(), output_type="latent",
                         ddim_init_latents_t_idx=0, ddim_inv_latents_path=traj, ddim_inv_prompt_embeds=g(ehs[:1]),
                         ddim_inv_image_embeddings=g(ie[:1]), ddim_inv_image_latents=g(il[:1]), latents_trace=ed)
    pnp_utils.clear_time(pipe)
    inv_ts = sorted(traj32.keys())
    fwd_ts = sorted(ed32.keys(), reverse=True)
    tag = f"{cfg_name} [{Fr}f x {hw * 8}^2]"
    rows = []
    for name, order, hip, o32_, o16_ in (("inversion", inv_ts, traj, traj32, traj16), ("reconstruction (inversion -> CFG sampling)", fwd_ts, rec, rec32, rec16),
                                         ("PnP edit (inversion -> sample_with_pnp)", fwd_ts, ed, ed32, ed16)):
        for i in range(every - 1, n_steps, every):
            t = order[i]
            # every row is enforced (round 2: only the final latent): bound = min(2 x eager-fp16 drift, 3 x the HIP drift on record)
            rows.append(_calibrated(f"{tag} {name}: drift after {i + 1} steps (t={t})", hip[t], o32_[t], o16_[t], floor=1e-3,
                                    key=f"drift:{cfg_name}:F{Fr}x{hw}:n{n_steps}:{name.split()[0]}:{i + 1}"))
    out.extend(rows)
    # the round trip itself (how well n-step reconstruction returns to the clean latent) -- same for all three paths
    for nm, r_ in (("HIP", rec[fwd_ts[-1]]), ("fp32 oracle", rec32[fwd_ts[-1]]), ("eager fp16 oracle", rec16[fwd_ts[-1]])):
        e = _res(f"{tag} round trip: {nm} reconstruction vs the clean latent", r_.float().cpu(), lat0.float(), 10.0)
        e["informational"] = True
        out.append(e)
    if report is not None:
        report["drift"] = [(r["name"], r["err"], r.get("eager")) for r in rows]
    return out


def check_inversion_500(cfg_name="full", Fr=16, hw=64, n_steps=500, every=100, report=None):
    """BASELINE config 2 as the reference ships it (configs/group_ddim_inversion/template.yaml:33: 500 inversion steps,
    timesteps 1, 3, ..., 999): the product pipeline's ``invert`` (HIP graphs) vs the oracle's ``invert_loop`` in fp32 on the GPU,
    latent drift every ``every`` steps, bounded by 2 x the eager-fp16 oracle's drift through the same loop."""
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from anyv2v_amd.schedulers import DDIMInverseScheduler
    from oracle import pnp_oracle
    out = []
    m = full_models(cfg_name, 1234, want=("native", "o32", "o16"))
    native, ocfg = m["native"], m["ocfg"]
    inp = config1_inputs(ocfg, 3, Fr, hw)
    h = lambda x: x.half()
    lat0 = h(inp["sample"][:1])
    ehs, ie, il = h(inp["encoder_hidden_states"]), h(inp["image_embeddings"]), h(inp["image_latents"])

    def oracle_run(o, dt):
        d = lambda x: x.to(DEV, dt)
        cond_src = dict(fps=torch.tensor([8], device=DEV), image_latents=d(il[:1]), image_embeddings=d(ie[:1]),
                        encoder_hidden_states=d(ehs[:1]))
        pnp_oracle.clear_hooks(o)
        return pnp_oracle.invert_loop(o, d(lat0), cond_src, n_steps)

    traj32 = {t: v.cpu() for t, v in oracle_run(m["o32"], torch.float32).items()}
    traj16 = {t: v.cpu() for t, v in oracle_run(m["o16"], torch.float16).items()}
    g = lambda x: x.to(DEV)
    pipe = I2VGenXLPipeline(unet=native, scheduler=DDIMInverseScheduler())
    pipe._device = torch.device(DEV)
    traj = pipe.invert(prompt_embeds=g(ehs[:1]), image_embeddings=g(ie[:1]), image_latents=g(il[:1]), height=hw * 8, width=hw * 8,
                       num_frames=Fr, num_inference_steps=n_steps, guidance_scale=1.0, target_fps=8, latents=g(lat0),
                       return_trajectory=True)
    ts = sorted(traj32.keys())
    assert len(ts) == n_steps and sorted(traj.keys()) == ts
    for i in range(every - 1, n_steps, every):
        t = ts[i]
        out.append(_calibrated(f"{cfg_name} [{Fr}f x {hw * 8}^2] {n_steps}-step inversion: drift after {i + 1} steps (t={t})", traj[t],
                               traj32[t], traj16[t], floor=1e-3, key=f"inv{n_steps}:{cfg_name}:F{Fr}x{hw}:{i + 1}"))
    if report is not None:
        report["inversion_500"] = [(r["name"], r["err"], r.get("eager")) for r in out]
    return out


def check_pipeline_vs_reference_fixture(name="mini"):
    """The product pipeline (HIP kernels + HIP graphs on the GPU; op emulation in the CPU suite) vs a fixture produced by the
    REFERENCE'S OWN PIPELINE CLASS on the CPU in fp32 (``tests/golden/make_golden.py --pipeline[-full]``: pipeline_i2vgen_xl.py
    verbatim around the oracle UNet): n-step inversion (every trajectory latent), CFG reconstruction and PnP edit from the
    reference's noisiest latent, on the conditioning tensors the reference's glue code built.  On the GPU every bound is 2 x the
    error of the same oracle model run through the oracle loops in eager fp16 on this GPU; on the CPU emulation a fixed bound."""
    from anyv2v_amd import pnp_utils
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    from oracle import pnp_oracle
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "ref_pipeline_mini.pt" if name == "mini" else "ref_pipeline_full_config1.pt"))
    sp = fx["spec"]
    calibrate = DEV != "cpu"
    m = full_models(sp["cfg"], sp["seed"], want=("native",) + (("o16",) if calibrate else ()))
    native, o16 = m["native"], m.get("o16")
    n_steps, size, Fr, ratios = sp["n_steps"], sp["size"], sp["frames"], sp["ratios"]
    g = lambda x: x.to(DEV)
    inv_ts, T = fx["inv_ts"], fx["T"]
    traj_ref = {t: fx["trajectory"][i] for i, t in enumerate(inv_ts)}
    pipe = I2VGenXLPipeline(unet=native, scheduler=DDIMInverseScheduler())
    pipe._device = torch.device(DEV)
    traj = pipe.invert(prompt_embeds=g(fx["src_pe"]), image_embeddings=g(fx["src_ie"]), image_latents=g(fx["src_il"]), height=size,
                       width=size, num_frames=Fr, num_inference_steps=n_steps, guidance_scale=1.0, target_fps=8, latents=g(fx["lat0"]),
                       return_trajectory=True)
    sched = DDIMScheduler()
    sched.set_timesteps(n_steps)
    pipe.register_modules(scheduler=sched)
    start = g(traj_ref[T])  # both sampling runs start from the REFERENCE's noisiest latent, as its own stage 2 does
    rec = pipe(prompt_embeds=g(fx["rec_pe"]), negative_prompt_embeds=g(fx["rec_npe"]), image_embeddings=g(fx["src_ie"]),
               image_latents=g(fx["src_il"]), height=size, width=size, num_frames=Fr, num_inference_steps=n_steps, guidance_scale=9.0,
               target_fps=8, latents=start.clone(), output_type="latent", ddim_init_latents_t_idx=0).frames
    k = lambda r: sched.timesteps[: int(n_steps * r)]
    pnp_utils.register_conv_injection(pipe, k(ratios[0]))
    pnp_utils.register_spatial_attention_pnp(pipe, k(ratios[1]))
    pnp_utils.register_temp_attention_pnp(pipe, k(ratios[2]))
    from anyv2v_amd.utils import LatentTrajectory
    src = LatentTrajectory()
    for t in inv_ts:
        src[t] = g(traj_ref[t])
    ed = pipe.sample_with_pnp(prompt_embeds=g(fx["pe"]), negative_prompt_embeds=g(fx["npe"]), image_embeddings=g(fx["ie_pos"]),
                              image_latents=g(fx["il_edit"]), height=size, width=size, num_frames=Fr, num_inference_steps=n_steps,
                              guidance_scale=9.0, target_fps=8, latents=start.clone(), output_type="latent",
                              ddim_init_latents_t_idx=0, ddim_inv_latents_path=src, ddim_inv_prompt_embeds=g(fx["src_pe"]),
                              ddim_inv_image_embeddings=g(fx["src_ie"]), ddim_inv_image_latents=g(fx["src_il"])).frames
    pnp_utils.clear_time(pipe)
    tag = f"{name} [{Fr}f x {size}^2, {n_steps} steps]"
    rows = [(f"{tag} pipe.invert vs the reference pipeline's trajectory, t={t}", traj[t], traj_ref[t]) for t in inv_ts]
    rows += [(f"{tag} pipe.__call__ (CFG reconstruction) vs the reference pipeline", rec, fx["rec_ref"]),
             (f"{tag} pipe.sample_with_pnp vs the reference pipeline", ed, fx["edit_ref"])]
    if not calibrate:
        return [_res(n_, a, b.float(), 5e-2) for n_, a, b in rows]
    # eager-fp16 calibration: the same oracle model through the oracle loops on this GPU, same inputs
    d = lambda x: x.to(DEV, torch.float16)
    with torch.no_grad():
        c1 = dict(fps=torch.tensor([8], device=DEV), image_latents=d(fx["src_il"]), image_embeddings=d(fx["src_ie"]),
                  encoder_hidden_states=d(fx["src_pe"]))
        t16 = pnp_oracle.invert_loop(o16, d(fx["lat0"]), c1, n_steps)
        c2 = dict(fps=torch.tensor([8, 8], device=DEV), image_latents=d(torch.cat([fx["src_il"]] * 2)),
                  image_embeddings=d(torch.cat([torch.zeros_like(fx["src_ie"]), fx["src_ie"]])),
                  encoder_hidden_states=d(torch.cat([fx["rec_npe"], fx["rec_pe"]])))
        r16 = pnp_oracle.sample_loop(o16, d(traj_ref[T]).clone(), c2, n_steps, 9.0, t_idx=0)
        c3 = dict(fps=torch.tensor([8, 8, 8], device=DEV), image_latents=d(torch.cat([fx["src_il"], fx["il_edit"], fx["il_edit"]])),
                  image_embeddings=d(torch.cat([fx["src_ie"], torch.zeros_like(fx["ie_pos"]), fx["ie_pos"]])),
                  encoder_hidden_states=d(torch.cat([fx["src_pe"], fx["npe"], fx["pe"]])))
        pnp_oracle.init_pnp(o16, n_steps, *ratios)
        e16 = pnp_oracle.pnp_loop(o16, d(traj_ref[T]).clone(), {t: d(v) for t, v in traj_ref.items()}, c3, n_steps, 9.0, t_idx=0)
        pnp_oracle.clear_hooks(o16)
    eager = [t16[t] for t in inv_ts] + [r16, e16]
    return [_calibrated(n_, a, b, e_, floor=1e-3) for (n_, a, b), e_ in zip(rows, eager)]


def check_full_size_properties():
    """BASELINE config 3 sizes (T = 196608 tokens, N = 48 images, S = 4096), where the CPU oracle is far too slow:
    size-independent identities of each kernel family."""
    out = []
    T, C = 196608, 320
    g = torch.Generator(device="cpu").manual_seed(7)
    r = lambda *sh, scale=1.0: (torch.randn(*sh, generator=g) * scale).to(torch.float16).to(DEV)
    # attention: softmax rows sum to one -> with V == 1 every output is exactly 1; key order does not matter
    N, h, S = 48, 5, 4096
    qk = r(N * S, 2 * C)
    ones = torch.ones(N * S, C, dtype=torch.float16, device=DEV)
    o = torch.empty(N * S, C, dtype=torch.float16, device=DEV)
    kw = dict(batch=N, heads=h, Sq=S, Sk=S, inner=1, q_strides=(S, 0, 1), kv_strides=(S, 0, 1))
    ops.attention(qk[:, :C], qk[:, C:], ones, o, **kw)
    out.append(_res("attention full size: V == 1 -> O == 1 (plain kernel)", o, ones.float(), 2e-3))
    ops.attention(qk[:, :C], qk[:, C:], ones, o, qk_mod=N // 3, **kw)
    out.append(_res("attention full size: V == 1 -> O == 1 (shared-softmax PnP kernel)", o, ones.float(), 2e-3))
    v = r(N * S, C)
    o1 = torch.empty_like(o)
    ops.attention(qk[:, :C], qk[:, C:], v, o1, **kw)
    perm = torch.randperm(S, generator=g).to(DEV)
    idx = (torch.arange(N, device=DEV)[:, None] * S + perm[None, :]).reshape(-1)
    o2 = torch.empty_like(o)
    ops.attention(qk[:, :C], qk[idx, C:].contiguous(), v[idx].contiguous(), o2, **kw)
    out.append(_res("attention full size: invariant under a permutation of the keys", o2, o1.float(), KTOL))
    # GEMM / conv linearity in the activations (fp32 accumulate, one rounding): f(a) + f(b) ~= f(a + b) with exact inputs
    a = (torch.randint(-8, 9, (T, C), generator=g).float() / 8).to(torch.float16).to(DEV)   # sums exact in fp16
    b = (torch.randint(-8, 9, (T, C), generator=g).float() / 8).to(torch.float16).to(DEV)
    w = (torch.randint(-4, 5, (C, C), generator=g).float() / 64).to(torch.float16).to(DEV)
    ya, yb, yab = ops.gemm(a, w), ops.gemm(b, w), ops.gemm((a + b), w)
    out.append(_res("linear 320->320 full size: f(a) + f(b) == f(a + b) on exactly representable data", yab, ya.float() + yb.float(), 1e-3))
    w9 = (torch.randint(-2, 3, (C, 9 * C), generator=g).float() / 64).to(torch.float16).to(DEV)
    cv = dict(mode=ops.MODE_CONV2D, conv=(64, 64, 64, 64, 1, 0))
    ca, cb, cab = ops.gemm(a, w9, **cv), ops.gemm(b, w9, **cv), ops.gemm((a + b), w9, **cv)
    out.append(_res("conv3x3 320->320 @64x64 full size: additive in the input", cab, ca.float() + cb.float(), 2e-3))
    # conv: shifting every image by one pixel row shifts the output (interior rows), i.e. the gather indexes correctly
    x4 = a.view(48, 64, 64, C)
    xs = torch.zeros_like(x4)
    xs[:, 1:] = x4[:, :-1]
    cs = ops.gemm(xs.reshape(T, C).contiguous(), w9, **cv).view(48, 64, 64, C)
    out.append(_res("conv3x3 full size: translation equivariance (rows 2..62)", cs[:, 3:62], ca.view(48, 64, 64, C)[:, 2:61].float(), 1e-3))
    # GroupNorm: per (frame, group) the normalised output has mean 0 / variance 1; LayerNorm likewise per row
    x = r(T, C) * 3 + 1
    stats = torch.empty(ops.gn_scratch_floats(48, 1), dtype=torch.float32, device=DEV)
    one, zero = torch.ones(C, dtype=torch.float16, device=DEV), torch.zeros(C, dtype=torch.float16, device=DEV)
    y = ops.groupnorm(x, one, zero, stats, 4096, groups=32, eps=1e-5).float().view(48, 4096, 32, 10)
    m, v_ = y.mean((1, 3)), y.var((1, 3), unbiased=False)
    out.append(_res("groupnorm full size: group means == 0", m + 1, torch.ones_like(m), 2e-3))
    out.append(_res("groupnorm full size: group variances == 1", v_, torch.ones_like(v_), KTOL))
    yl = ops.layernorm(x, one, zero, 1e-5).float()
    out.append(_res("layernorm full size: row means == 0", yl.mean(1) + 1, torch.ones(T, device=DEV), 2e-3))
    out.append(_res("layernorm full size: row variances == 1", yl.var(1, unbiased=False), torch.ones(T, device=DEV), KTOL))
    return out


def check_vae_kernels():
    """The kernel modes added for the AutoencoderKL: one-sided-pad stride-2 conv, fp32-output GEMM, row softmax, and the
    logits -> softmax -> value chain as one 512-wide attention head."""
    out = []
    for naive in (False, True):
        tag = "naive" if naive else "mfma"
        n, ci, co, H, W = 4, 64, 128, 16, 12
        x, w, b = rnd(n, ci, H, W), rnd(co, ci, 3, 3, scale=1 / math.sqrt(9 * ci)), rnd(co)
        y = ops.gemm(_to_tokens(x), _pack_conv(w), bias=b, mode=ops.MODE_CONV2D, conv=(H, W, H // 2, W // 2, 2, 0, 1),
                     M=n * (H // 2) * (W // 2), naive=naive)
        ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b.float(), stride=2)
        out.append(_res(f"conv3x3[{tag}] stride 2, pad (0,1,0,1) (VAE Downsample2D)", y, _to_tokens(ref), KTOL))
        a, wk = rnd(300, 128), rnd(256, 128, scale=1 / math.sqrt(128))
        bias = rnd(256)
        s32 = ops.gemm(a, wk, bias=bias, act=ops.ACT_F32OUT, naive=naive)
        ref32 = a.float() @ wk.float().t() + bias.float()
        out.append(_res(f"gemm[{tag}] fp32 output", s32, ref32, 2e-5))
    s32 = torch.randn(500, 1000, device=DEV) * 30
    p = ops.softmax_rows(s32, 0.0442)
    out.append(_res("softmax rows fp32 -> fp16", p, torch.softmax(s32 * 0.0442, -1), 2e-3))
    out.append(_res("softmax rows sum to 1", p.float().sum(-1), torch.ones(500, device=DEV), 2e-3))
    # single head, d = 512, S = 1024: logits GEMM (fp32) -> softmax -> value GEMM
    S, C = 1024, 512
    q, k, v = rnd(S, C), rnd(S, C), rnd(S, C)
    logits = ops.gemm(q, k, act=ops.ACT_F32OUT)
    o = ops.gemm(ops.softmax_rows(logits, C ** -0.5), v.t().contiguous())
    ref = F.scaled_dot_product_attention(q.float()[None, None], k.float()[None, None], v.float()[None, None])[0, 0]
    out.append(_res("single 512-wide attention head via GEMM / softmax / GEMM", o, ref, KTOL))
    return out


def check_vae(full: bool = True):
    """Native AutoencoderKL vs the CPU oracle on identical fp16-rounded random weights: mini config, and the full SD-VAE
    architecture (83.65 M parameters) on 2 x 64x64 images / 2 x 8x8 latents."""
    from anyv2v_amd.vae import AutoencoderKL, VAEConfig
    from oracle import vae_oracle as vo
    out = []
    cases = [("mini", VAEConfig.mini(), vo.VAEConfig.mini(), (2, 3, 32, 48), (2, 4, 8, 12))]
    if full:
        cases.append(("full", VAEConfig(), vo.VAEConfig(), (2, 3, 64, 64), (2, 4, 8, 8)))
    for name, ncfg, ocfg, xs, zs in cases:
        oracle = vo.AutoencoderKLOracle(ocfg)
        sd = vo.random_state_dict(ocfg, 11)
        oracle.load_state_dict(sd)
        native = AutoencoderKL(ncfg)
        native.load_state_dict(sd)
        native.to(DEV)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(*xs, generator=g).clamp(-1, 1).half().float()
        z = torch.randn(*zs, generator=g).half().float()
        m0, l0 = oracle.encode_moments(x)
        m1, l1 = native.encode_moments(x.to(DEV))
        # bounds = 3 x the measured error of the full-width model on an MI355X (round 4: mean 1.6e-3, logvar 1.2e-3, decode 2.3e-3)
        out.append(_res(f"vae[{name}] encode: posterior mean vs oracle", m1.cpu(), m0, 5e-3))
        out.append(_res(f"vae[{name}] encode: posterior logvar vs oracle", l1.cpu(), l0, 4e-3))
        d0 = oracle.decode(z)
        d1 = native.decode(z.to(DEV))
        out.append(_res(f"vae[{name}] decode vs oracle", d1.cpu(), d0, 7e-3))
    return out


def check_vae_blocks_vs_reference_fixture():
    """The NATIVE VAE blocks (``anyv2v_amd/vae.py``: ``VAEResnetBlock``, the folded nearest-x2 up-sampler, the one-sided-pad stride-2
    down-sampler) on the HIP kernels against ``tests/golden/vae_blocks_ref.pt`` -- outputs of the reference's vendored copies of those
    diffusers blocks (``seine/models/resnet.py:24-207``, ``make_golden.py --vae-blocks``).  Row F1's pinnable part."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as mg
    from anyv2v_amd import vae as nv
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "vae_blocks_ref.pt"))
    spec = fx["spec"]
    io = mg.vae_block_inputs(spec)
    H, W = spec["hw"]
    sc = nv._Scratch()
    out = []

    def back(t, n, h, w):
        return t.float().view(n, h, w, -1).permute(0, 3, 1, 2).cpu()
    for cin, cout in spec["cases"]:
        name = f"res{cin}_{cout}"
        blk = nv.VAEResnetBlock(cin, cout, spec["groups"])
        blk.load_state_dict(io["weights"][name])
        blk = blk.to(DEV).half()
        for m in blk.modules():
            if hasattr(m, "pack"):
                m.pack()
        y = blk.run(sc, _to_tokens(io["x"][name].to(DEV).half()), H, W)
        out.append(_res(f"vae block {name} (native) vs the reference's ResnetBlock3D(temb=None) fixture", back(y, spec["n"], H, W), fx["out"][name], 3e-3))
    c = spec["sampler_c"]
    for name, stride in (("up", 1), ("down", 2)):
        smp = nv._Sampler(c, stride)
        smp.load_state_dict(io["weights"][name])
        smp = smp.to(DEV).half()
        for m in smp.modules():
            if hasattr(m, "pack"):
                m.pack()
        x = _to_tokens(io["x"][name].to(DEV).half())
        if name == "up":
            y = smp.conv.tokens(x, H, W, up=True)
            out.append(_res("vae up-sampler (native, nearest x2 folded) vs the reference's Upsample3D fixture", back(y, spec["n"], 2 * H, 2 * W), fx["out"]["up"], 3e-3))
        else:
            y = smp.conv.tokens(x, H, W, asym=True)
            out.append(_res("vae down-sampler (native, pad (0,1,0,1) stride 2) vs the reference's Downsample3D fixture (shifted-input identity)",
                            back(y, spec["n"], H // 2, W // 2), fx["out"]["down"], 3e-3))
    return out


def check_consisti2v_hooks():
    """the ConsistI2V hook family (``anyv2v_amd/consisti2v.py``) on the kernels vs the fixture the REFERENCE's
    own ``VideoLDMCrossAttnUpBlock`` + ``consisti2v/pnp_utils.py`` produced on the CPU in fp32 (``make_golden.py --consisti2v``):
    stand-ins for ``unet.up_blocks[1..3]``, un-hooked and with conv + spatial + temporal injection (t = 981) / temporal only (301)."""
    import consisti2v_spec as spec
    from anyv2v_amd import consisti2v as c2
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "consisti2v_decoder_hooks.pt"))
    blocks = {i: spec.fill_weights(c2.VideoLDMCrossAttnUpBlock(**spec.block_kwargs(i))).to(DEV) for i in spec.BLOCKS}

    def call(blk, x, skips, temb, ehs):
        h = lambda t: t.to(DEV).half()
        return blk(h(x), tuple(h(s) for s in skips), h(temb), encoder_hidden_states=h(ehs)).float().cpu()
    got = spec.run_cases(blocks, c2, call)
    out = []
    for i in spec.BLOCKS:
        for case in ["nohook"] + [f"hook_t{t}" for t in spec.TS_CASES]:
            out.append(_res(f"consisti2v up_blocks[{i}] stand-in, {case} vs the reference's own block + hooks", got[f"block{i}_{case}"],
                            fx[f"block{i}_{case}"], 4e-3))
        out.append(dict(name=f"consisti2v up_blocks[{i}]: a timestep outside every schedule == un-hooked (bit-equal)", err=0.0, tol=0.0,
                        ok=bool(torch.equal(got[f"block{i}_nohook"], got[f"block{i}_hook_t101"]))))
    return out


def check_consisti2v_unet():
    """The whole ConsistI2V UNet (``anyv2v_amd/consisti2v.py:VideoLDMUNet3DConditionModel``) on the kernels vs the fixture the
    REFERENCE's own ``VideoLDMUNet3DConditionModel`` + ``consisti2v/pnp_utils.py`` produced on the CPU in fp32
    (``make_golden.py --consisti2v-unet``): toy width, every block type, first-frame concatenation, frame-stride conditioning."""
    import consisti2v_spec as spec
    from anyv2v_amd import consisti2v as c2
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "consisti2v_unet.pt"))
    unet = spec.fill_weights(c2.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)).to(DEV)

    def call(u, sample, t, ehs, first, stride):
        h = lambda v: v.to(DEV).half()
        return u(h(sample), t, encoder_hidden_states=h(ehs), first_frame_latents=h(first), frame_stride=stride).sample.float().cpu()
    got = spec.run_unet_cases(unet, c2, call)
    out = []
    for case in ["nohook"] + [f"hook_t{t}" for t in spec.TS_CASES]:
        out.append(_res(f"consisti2v whole UNet (toy width), {case} vs the reference's own UNet + hooks", got[f"unet_{case}"],
                        fx[f"unet_{case}"], 8e-3))
    out.append(dict(name="consisti2v whole UNet: a timestep outside every schedule == un-hooked (bit-equal)", err=0.0, tol=0.0,
                    ok=bool(torch.equal(got["unet_nohook_t101"], got["unet_hook_t101"]))))
    return out


def check_consisti2v_unet_full():
    """ConsistI2V at the released model's width (1250 M parameters, 16 frames x 32 x 32 latent pixels, [source, negative, editing]) on
    the kernels vs the fixture the REFERENCE's own ``VideoLDMUNet3DConditionModel`` + hooks produced on the CPU in fp32
    (``make_golden.py --consisti2v-unet-full``): the flash kernel with Sk = 2 HW, the whole-sequence MFMA kernel at head_dim 40 / 80 /
    160 with 16 + 8 keys, the fused feed-forward, the persistent GEMM kernels -- everything the timing tool runs."""
    import consisti2v_spec as spec
    from anyv2v_amd import consisti2v as c2
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "consisti2v_unet_full.pt"))
    unet = spec.fill_weights(c2.VideoLDMUNet3DConditionModel(**spec.unet_full_cfg())).to(DEV)

    def call(u, sample, t, ehs, first, stride):
        h = lambda v: v.to(DEV).half()
        return u(h(sample), t, encoder_hidden_states=h(ehs), first_frame_latents=h(first), frame_stride=stride).sample.float().cpu()
    got = spec.run_unet_full_cases(unet, c2, call)
    out = []
    for case in ("full_nohook", "full_hook_t981"):
        out.append(_res(f"consisti2v UNet at the released width (1250 M), {case} vs the reference's own UNet + hooks (fp32 CPU)",
                        got[case], fx[case].float(), 1e-2))
    # (not bit-equal at this size: under injection the source branch's ResNet main path and attention run as launches of their own,
    # a third of the rows, which take other tile / split-K plans -- rounding-level)
    out.append(_res("consisti2v full width: the source branch with hooks on vs off", got["full_hook_t981"][:1], got["full_nohook"][:1], 5e-3))   # measured 1.95e-3 (each is 2.4-2.7e-3 from fp32)
    del unet
    torch.cuda.empty_cache()
    return out


def check_consisti2v_pipeline():
    """ConsistI2V end to end, pipeline level: ``anyv2v_amd.consisti2v_pipeline.ConditionalVideoEditingPipeline`` on the kernels --
    ``encode_vae_video``, ``invert``, ``__call__`` (reconstruction), ``sample_with_pnp`` -- vs the fixture the REFERENCE's own pipeline
    class produced on the CPU in fp32 (``make_golden.py --consisti2v-pipeline``, ``oracle/ref_consisti2v_pipeline.py``); every stage
    starts from the reference's own trajectory.  The edit row (text guidance 35) is gated against an independent fp16-storage run, not a constant."""
    import consisti2v_spec as spec
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "consisti2v_pipeline.pt"))
    files = {t: fx["trajectory"][i] for i, t in enumerate(fx["inv_ts"])}
    nat = spec.native_pipeline_job(DEV, trajectory_from=files)
    out = [dict(name="consisti2v pipeline: inversion timesteps", err=0.0, tol=0.0, ok=nat["inv_ts"] == fx["inv_ts"])]
    out.append(_res("consisti2v pipeline: encode_vae_video", nat["lat0"].float().cpu(), fx["lat0"].float(), 3e-3))
    for i, t in enumerate(fx["inv_ts"]):
        out.append(_res(f"consisti2v pipeline: invert, latents written at t={t}", nat["files"][t].float().cpu(), fx["trajectory"][i].float(), 2e-2))
    out.append(_res("consisti2v pipeline: __call__ reconstruction from t_idx 1", nat["rec_lat"].float().cpu(), fx["rec_lat"].float(), 2e-2))
    # no fixed bound on the amplified row.  Text guidance 35 multiplies the difference of two branch predictions --
    # and their fp16 rounding -- by 35; the bound is 2 x what an independent fp16-storage implementation (the torch op emulation, run
    # here on the CPU on the same weights from the same trajectory) shows against the reference's fp32 output, and 3 x the recorded
    # HIP figure; HIP-vs-emulation is printed.
    emu = _emulated(lambda: spec.native_pipeline_job("cpu", trajectory_from=files))
    # (the emulation arm runs the repo's own model / pipeline code -- it is independent at KERNEL level only -- so it may
    #  tighten but never loosen the former fixed bound of 0.25, and the HIP-vs-emulation distance is capped at 1.5 x its recorded 0.19)
    out.append(_calibrated("consisti2v pipeline: sample_with_pnp (text guidance 35) vs the reference class's output; 'eager' = torch op emulation "
                           "(kernel-independent only)", nat["edit_lat"], fx["edit_lat"], emu["edit_lat"], key="consisti2v_pipeline_pnp_edit",
                           max_tol=0.25, gap_cap=0.29))
    out.append(_calibrated("consisti2v pipeline: __call__ reconstruction (calibrated)", nat["rec_lat"], fx["rec_lat"], emu["rec_lat"],
                           key="consisti2v_pipeline_reconstruction"))
    dec = torch.from_numpy(nat["pipe"].decode_latents(fx["edit_lat"].to(DEV)))
    out.append(_res("consisti2v pipeline: decode_latents of the reference's edited latents", dec, fx["edit_video"].float(), 4e-3))
    return out


def check_consisti2v_sampling():
    """ConsistI2V's samplers next to the runner stages -- ``ConditionalAnimationPipeline``, ``AutoregressiveAnimationPipeline`` (two
    chunks) and ``guidance_rescale`` + ``eta`` on the editing pipeline -- on the kernels vs the fixture the REFERENCE's own classes
    produced on the CPU in fp32 (``make_golden.py --consisti2v-sampling``); seeded noise, drawn on the host on both sides."""
    import consisti2v_spec as spec
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "consisti2v_sampling.pt"))
    got = spec.native_sampling(DEV)
    tol = dict(animation=8e-3, autoregressive=8e-3, rescale_eta=2e-2)      # (3 x measured, profiles/r04_consisti2v_sampling_gpu.txt)
    return [_res(f"consisti2v sampling: {name} ({spec.sampling_cases()[name][0]})", lat.float().cpu(), fx[name].float(), tol[name])
            for name, lat in got.items()]


def check_seine_hooks():
    """the SEINE hook family (``anyv2v_amd/seine.py``) on the kernels vs the fixture the REFERENCE's own
    ``CrossAttnUpBlock3D`` + ``seine/pnp_utils.py`` produced on the CPU in fp32 (``make_golden.py --seine``): un-hooked, and with
    conv + spatial + cross + temporal injection (t = 981) / cross + temporal (501) / temporal only (301)."""
    import seine_spec as spec
    from anyv2v_amd import seine as sn
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "seine_decoder_hooks.pt"))
    blocks = {i: spec.fill_weights(sn.CrossAttnUpBlock3D(**spec.block_kwargs(i)), spec.WEIGHT_SEED).to(DEV) for i in spec.BLOCKS}

    def call(blk, x, skips, temb, ehs):
        h = lambda t: t.to(DEV).half()
        return blk(h(x), tuple(h(s) for s in skips), h(temb), encoder_hidden_states=h(ehs)).float().cpu()
    got = spec.run_cases(blocks, sn, call)
    out = []
    for i in spec.BLOCKS:
        for case in ["nohook"] + [f"hook_t{t}" for t in spec.TS_CASES]:
            out.append(_res(f"seine up_blocks[{i}] stand-in, {case} vs the reference's own block + hooks", got[f"block{i}_{case}"],
                            fx[f"block{i}_{case}"], 4e-3))
        out.append(dict(name=f"seine up_blocks[{i}]: a timestep outside every schedule == un-hooked (bit-equal)", err=0.0, tol=0.0,
                        ok=bool(torch.equal(got[f"block{i}_nohook"], got[f"block{i}_hook_t101"]))))
    return out


def check_seine_unet():
    """The whole SEINE UNet (``anyv2v_amd/seine.py:UNet3DConditionModel``) on the kernels vs the fixture the REFERENCE's own
    ``UNet3DConditionModel`` + ``seine/pnp_utils.py`` produced on the CPU in fp32 (``make_golden.py --seine-unet``)."""
    import seine_spec as spec
    from anyv2v_amd import seine as sn
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "seine_unet.pt"))
    unet = spec.fill_weights(sn.UNet3DConditionModel(**spec.UNET_CFG), spec.WEIGHT_SEED).to(DEV)

    def call(u, sample, t, ehs):
        return u(sample.to(DEV).half(), t, encoder_hidden_states=ehs.to(DEV).half()).sample.float().cpu()
    got = spec.run_unet_cases(unet, sn, call)
    out = []
    for case in ["nohook"] + [f"hook_t{t}" for t in spec.TS_CASES]:
        out.append(_res(f"seine whole UNet (toy width), {case} vs the reference's own UNet + hooks", got[f"unet_{case}"], fx[f"unet_{case}"], 8e-3))
    out.append(dict(name="seine whole UNet: a timestep outside every schedule == un-hooked (bit-equal)", err=0.0, tol=0.0,
                    ok=bool(torch.equal(got["unet_nohook_t101"], got["unet_hook_t101"]))))
    return out


def check_seine_pipeline():
    """SEINE end to end, runner-class level: ``anyv2v_amd.seine_pipeline`` on the kernels vs the fixture the REFERENCE's own
    ``SEINEDDIMInversionPipeline`` / ``SEINEPnPPipeline`` produced on the CPU (``make_golden.py --seine-pipeline``), DDIM and DDPM samplers
    (the DDPM noise is drawn from the CPU generator, the fixture's stream); the edit starts from the reference's own trajectory."""
    import tempfile

    import seine_spec as spec
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "seine_pipeline.pt"))
    files = {t: fx["trajectory"][i] for i, t in enumerate(fx["inv_ts"])}
    out = []
    for sm in ("ddim", "ddpm"):
        with tempfile.TemporaryDirectory() as tmp:
            nat = spec.native_job(DEV, tmp, sm, trajectory_from=files)
        if sm == "ddim":
            out.append(_res("seine runners: frames -> VAE latents", nat["lat0"].float().cpu(), fx["lat0"].float(), 2e-3))
            for i, t in enumerate(fx["inv_ts"]):
                out.append(_res(f"seine runners: ddim_inversion, file at t={t}", nat["files"][t].float().cpu(), fx["trajectory"][i].float(), 1e-2))
            out.append(_res("seine runners: ddim_sample reconstruction", nat["recon_lat"].float().cpu(), fx["recon_lat"].float(), 2e-2))
        out.append(dict(name=f"seine runners: edit timesteps ({sm})", err=0.0, tol=0.0, ok=nat["edit_ts"] == fx[f"edit_ts_{sm}"]))
        with tempfile.TemporaryDirectory() as tmp2:   # (calibrated like the ConsistI2V edit row)
            emu = _emulated(lambda: spec.native_job("cpu", tmp2, sm, trajectory_from=files))
        out.append(_calibrated(f"seine runners: edit_video, {sm} sampler, cfg 4 vs the reference runner's output; 'eager' = torch op emulation "
                               "(kernel-independent only)", nat["edit_lat"], fx[f"edit_lat_{sm}"], emu["edit_lat"], key=f"seine_pipeline_edit_{sm}",
                               max_tol=5e-2, gap_cap=4.5e-2))
        dec = nat["pipe"].decode_latents(fx[f"edit_lat_{sm}"].to(DEV))
        d = int((dec.int() - fx[f"edited_frames_{sm}"].int()).abs().max())
        out.append(dict(name=f"seine runners: decode_latents of the reference's latents ({sm}), max |uint8 diff|", err=float(d), tol=1.0, ok=d <= 1))
    return out


def check_attention_bias_and_rotary_windows():
    """``anyv2v_attention_bias_f16`` (additive score bias [heads, Sq, Sk], frame-strided sequences, qk_mod aliasing) and
    ``anyv2v_rotary_f16`` with one window per head vs PyTorch fp32."""
    out = []
    # (head_dim 40 / 80 / 160, 16 frames: SEINE's temporal attention at the released width -- the whole-sequence MFMA kernel with the
    # bias added to the score fragments; head_dim 20: the one-thread-per-query kernel)
    for (heads, D, Fr) in ((3, 40, 6), (8, 80, 16), (8, 160, 16), (2, 20, 5)):
        B, HW = 3, 10
        C = heads * D
        qkv = rnd(B * Fr * HW, 3 * C, seed=D)
        bias = torch.randn(heads, Fr, Fr, device=DEV)
        o = torch.zeros(B * Fr * HW, C, dtype=torch.float16, device=DEV)
        for qk_mod in (0, B * HW // 3):
            kw = dict(batch=B * HW, heads=heads, Sq=Fr, Sk=Fr, inner=HW, q_strides=(Fr * HW, 1, HW), kv_strides=(Fr * HW, 1, HW), qk_mod=qk_mod,
                      scale=D ** -0.5, head_dim=D, bias=bias)
            ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, **kw)
            x = qkv.float().view(B, Fr, HW, 3, heads, D).permute(3, 0, 2, 4, 1, 5)      # [3][B, HW, heads, F, D]
            q, k, v = x[0], x[1], x[2]
            if qk_mod:
                q, k = q[:1].expand_as(q), k[:1].expand_as(k)
            ref = F.scaled_dot_product_attention(q, k, v, attn_mask=bias[None, None], scale=D ** -0.5)   # [B, HW, heads, F, D]
            ref = ref.permute(0, 3, 1, 2, 4).reshape(B * Fr * HW, C)
            out.append(_res(f"attention + score bias, temporal view, {heads} x {D}, {Fr} frames, qk_mod {qk_mod}", o, ref, KTOL))
            o2 = torch.zeros_like(o)
            ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o2, naive=True, **kw)
            out.append(_res(f"attention + score bias == one-thread-per-query kernel, {heads} x {D}, qk_mod {qk_mod}", o, o2.float(), 1.5e-3))
    B, Fr, HW, heads, D = 3, 6, 10, 3, 40
    C = heads * D
    xr = rnd(2 * Fr * HW, C + 16)
    got = ops.rotary(xr.clone(), 8, 32, HW, Fr, windows=heads, window_stride=D)
    pos = ((torch.arange(xr.shape[0], device=DEV) // HW) % Fr).float()
    freq = 10000.0 ** (-torch.arange(0, 32, 2, device=DEV).float() / 32)
    ang = (pos[:, None] * freq[None]).repeat_interleave(2, dim=1)
    want = xr.float().clone()
    for hh in range(heads):
        c0 = 8 + hh * D
        t = xr[:, c0:c0 + 32].float()
        rot = torch.stack([-t[:, 1::2], t[:, 0::2]], -1).reshape(t.shape)
        want[:, c0:c0 + 32] = t * ang.cos() + rot * ang.sin()
    out.append(_res("rotary, one 32-channel window per head", got, want, 2e-3))
    return out


ALL_KERNEL_CHECKS = [check_selftest, check_gemm, check_gemm_big, check_gemm_ws, check_gemm_ws_ln, check_gemm_splitk, check_conv, check_norms, check_attention,
                     check_attention_small_mfma, check_attention_bias_and_rotary_windows, check_gelu_all_inputs, check_elementwise,
                     check_full_size_properties, check_vae_kernels]


def check_frame_parallel(Fr=4, hw=16, tol=4e-3):
    """Frame-parallel clip (``anyv2v_amd.parallel.FrameParallel``): called on every rank of an
    initialised process group; the sharded forward must reproduce this rank's own unsharded forward (same kernels, same
    weights; only the order of the fp32 additions inside the 5-D GroupNorm statistics differs), with and without the
    PnP hooks and the shared stem, and leave the pipeline untouched."""
    import types
    from anyv2v_amd import pnp_utils
    from anyv2v_amd.parallel import FrameParallel
    from oracle import pnp_oracle
    out = []
    fp = FrameParallel()
    native, oracle, ocfg = build_pair("mini", 1234)
    otol = 8e-3   # mini model, HIP fp16 (or the CPU emulation) vs the fp32 CPU oracle: 3 x the measured 2.0-2.4e-3 (round 4)
    for B in (1, 3):
        inp = config1_inputs(ocfg, B, Fr, hw)
        if B == 3:  # the edit loop's batch: slots 1 and 2 share latent and image latents (shared stem allowed)
            inp["sample"][2] = inp["sample"][1]
            inp["image_latents"][2] = inp["image_latents"][1]
        smp = inp["sample"].half().to(DEV)
        kw = dict(fps=inp["fps"].to(DEV), image_latents=inp["image_latents"].half().to(DEV),
                  image_embeddings=inp["image_embeddings"].half().to(DEV),
                  encoder_hidden_states=inp["encoder_hidden_states"].half().to(DEV))
        cases = [("no hooks", 981, False)]
        if B == 3:
            ts = [981 - 20 * i for i in range(50)]
            pipe = types.SimpleNamespace(unet=native)
            pnp_utils.register_conv_injection(pipe, ts[:10])
            pnp_utils.register_spatial_attention_pnp(pipe, ts[:25])
            pnp_utils.register_temp_attention_pnp(pipe, ts[:40])
            pnp_oracle.register_conv_injection(oracle, ts[:10])
            pnp_oracle.register_spatial_attention_pnp(oracle, ts[:25])
            pnp_oracle.register_temp_attention_pnp(oracle, ts[:40])
            cases = [("PnP all sites", 981, False), ("PnP temporal only + shared stem", 301, True), ("PnP off", 1, True)]
        for name, t, shared in cases:
            if B == 3:
                pnp_utils.register_time(pipe, t)
                pnp_oracle.register_time(oracle, t)
            with torch.no_grad():  # the checker: fp32 CPU oracle on the same fp16-rounded inputs (not only "vs itself")
                v_or = oracle(smp.float().cpu(), t, fps=inp["fps"], image_latents=inp["image_latents"].half().float(),
                              image_embeddings=inp["image_embeddings"].half().float(),
                              encoder_hidden_states=inp["encoder_hidden_states"].half().float())[0]
            res = []
            for use_fp in (None, fp):
                native.set_frame_parallel(use_fp)
                native.forward_tokens(smp, t, kw["fps"], kw["image_latents"], kw["image_embeddings"], kw["encoder_hidden_states"])
                native._ctx.shared_stem = shared
                res.append(native(smp, t, **kw)[0].float().cpu())
            out.append(_res(f"frame-parallel x{fp.world} rank {fp.rank}: B{B} {name} vs unsharded", res[1], res[0], tol))
            out.append(_res(f"frame-parallel x{fp.world} rank {fp.rank}: B{B} {name} vs the fp32 CPU oracle", res[1], v_or, otol))
        if B == 3:
            pnp_oracle.clear_hooks(oracle)
    # the pipeline loops on top (unchanged code: every rank steps the full, replicated latents): 4-step inversion, then
    # a 4-step PnP edit with schedules that end early (so the 2-branch steps are covered too)
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    inp = config1_inputs(ocfg, 3, Fr, hw)
    h = lambda x: x.half().to(DEV)
    lat0, ehs, ie, il = h(inp["sample"][:1]), h(inp["encoder_hidden_states"]), h(inp["image_embeddings"]), h(inp["image_latents"])
    n_steps, finals = 4, []
    for use_fp in (None, fp):
        native.set_frame_parallel(use_fp)
        pipe = I2VGenXLPipeline(unet=native, scheduler=DDIMInverseScheduler())
        pipe._device = torch.device(DEV)
        traj = pipe.invert(prompt_embeds=ehs[:1], image_embeddings=ie[:1], image_latents=il[:1], height=hw * 8, width=hw * 8,
                           num_frames=Fr, num_inference_steps=n_steps, guidance_scale=1.0, target_fps=8, latents=lat0,
                           return_trajectory=True)
        T = max(traj.keys())
        sched = DDIMScheduler()
        sched.set_timesteps(n_steps)
        pnp_utils.register_conv_injection(pipe, sched.timesteps[:1])
        pnp_utils.register_spatial_attention_pnp(pipe, sched.timesteps[:2])
        pnp_utils.register_temp_attention_pnp(pipe, sched.timesteps[:3])
        pipe.register_modules(scheduler=sched)
        res = pipe.sample_with_pnp(prompt_embeds=ehs[2:3], negative_prompt_embeds=ehs[1:2], image_embeddings=ie[2:3],
                                   image_latents=il[2:3], height=hw * 8, width=hw * 8, num_frames=Fr,
                                   num_inference_steps=n_steps, guidance_scale=9.0, target_fps=8, latents=traj[T].clone(),
                                   output_type="latent", ddim_init_latents_t_idx=0, ddim_inv_latents_path=traj,
                                   ddim_inv_prompt_embeds=ehs[:1], ddim_inv_image_embeddings=ie[:1],
                                   ddim_inv_image_latents=il[:1]).frames
        finals.append((traj[T].float().cpu(), res.float().cpu()))
        pnp_utils.clear_time(pipe)
    out.append(_res(f"frame-parallel rank {fp.rank}: pipeline.invert {n_steps} steps vs unsharded", finals[1][0], finals[0][0], 6e-3))   # measured 1.5e-3
    out.append(_res(f"frame-parallel rank {fp.rank}: pipeline.sample_with_pnp {n_steps} steps vs unsharded", finals[1][1],
                    finals[0][1], 4e-2))
    native.set_frame_parallel(None)
    out.append({"name": f"frame-parallel rank {fp.rank}: all-to-all payload > 0", "err": 0.0 if fp.bytes_moved > 0 else 1.0,
                "tol": 0.5, "ok": fp.bytes_moved > 0})
    return out
