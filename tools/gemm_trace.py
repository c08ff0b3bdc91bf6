"""In-kernel phase timeline of the 128-row GEMM kernel (GPU, debug flag bit5): where a block's time goes.

Each block's thread 0 records s_memtime at: start, first tile landed (prologue), after the MFMAs of K-tile k, after the
end-of-tile wait+barrier of K-tile k (first 8 tiles), end of K loop, end of epilogue (stores drained).
Writes gpurun_out/gemm_trace.txt.     python tools/gemm_trace.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from anyv2v_amd import _lib, ops  # noqa: E402

# probe build with the trace / knock-out instantiations (make -C anyv2v_amd/csrc experiments); never the product library
_lib.LIB_PATH = os.path.join(ROOT, "tools", "libanyv2v_hip_experiments.so")

dev = "cuda"
lines = []


def emit(s=""):
    lines.append(s)
    print(s, flush=True)


def run(tag, M, N, K, mode=0, act=0, conv=None, temporal=None, res=False, a_rows=None):
    taps = {0: 1, 1: 9, 2: 3}[mode]
    a = torch.randn(a_rows or M, K // taps, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device=dev)
    n_out = N // 2 if act == 3 else N
    out = torch.empty(M, n_out, dtype=torch.float16, device=dev)
    r = torch.randn(M, n_out, device=dev).half() if res else None
    kw = dict(bias=b, out=out, mode=mode, act=act, conv=conv, temporal=temporal, residual=r, M=M)
    ws = ops._workspace(torch.device(dev, 0))
    for _ in range(2):
        ops.gemm(a, w, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm(a, w, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    saved = ops.GEMM_FLAGS
    if act != 3 and N % 160 == 0:  # knock-outs: what the K loop costs without one of its ingredients
        ko_names = {2: "no A DMA", 3: "no DMA", 4: "no frag reads", 5: "no MFMA"}
        res_ko = []
        for ko, name in ko_names.items():
            ops.GEMM_FLAGS = saved | 4 | (ko << 6)
            for _ in range(2):
                ops.gemm(a, w, **kw)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                ops.gemm(a, w, **kw)
            e1.record()
            torch.cuda.synchronize()
            res_ko.append(f"{name} {e0.elapsed_time(e1) / 5 * 1e3:.1f} us")
        ops.GEMM_FLAGS = saved
        emit(f"{tag}: full {us:.1f} us | knock-outs: " + " | ".join(res_ko))
    ops.GEMM_FLAGS = saved | 4 | 32
    try:
        ws.zero_()
        ops.gemm(a, w, **kw)
        ops.gemm(a, w, **kw)
        torch.cuda.synchronize()
    finally:
        ops.GEMM_FLAGS = saved
    nf = 4 if act == 3 else (5 if N % 160 == 0 else 4)
    nblk = ((M + 127) // 128) * ((N + nf * 32 - 1) // (nf * 32))
    t = ws.view(torch.int64)[: nblk * 32].cpu().numpy().reshape(nblk, 32)
    nk = int(t[0, 23])
    assert nk > 0, "trace variant did not run"
    mhz = np.median((t[:, 21] - t[:, 2]) / ((t[:, 22] - t[:, 1]) / 100.0))  # memtime ticks per us
    ns = lambda ticks: ticks / mhz * 1e3
    span_us = (t[:, 22].max() - t[:, 1].min()) / 100.0
    tot = ns(t[:, 21] - t[:, 2])
    pro = ns(t[:, 3] - t[:, 2])
    epi = ns(t[:, 21] - t[:, 20])
    kk = min(nk, 8)
    mma = np.stack([ns(t[:, 4 + k] - (t[:, 3] if k == 0 else t[:, 12 + k - 1])) for k in range(kk)], 1)
    wait = np.stack([ns(t[:, 12 + k] - t[:, 4 + k]) for k in range(kk)], 1)
    loop = ns(t[:, 20] - t[:, 3])
    cu = (t[:, 0] >> 32) * 4096 + ((t[:, 0] & 0xFFFFFFFF) >> 8 & 0xF) + 16 * ((t[:, 0] & 0xFFFFFFFF) >> 13 & 0x7) + 128 * ((t[:, 0] & 0xFFFFFFFF) >> 12 & 1)
    ncu = len(np.unique(cu))
    conc = tot.sum() / 1e3 / span_us / ncu
    emit(f"{tag}: M={M} N={N} K={K} mode={mode} act={act} res={res}: {us:.1f} us untraced "
         f"({2.0 * M * N * K / us / 1e6:.0f} TF/s), traced span {span_us:.1f} us, {nblk} blocks on {ncu} CUs, "
         f"{nk} K-tiles, memtime {mhz:.0f} ticks/us")
    emit(f"    block total {tot.mean():8.0f} ns (p10 {np.percentile(tot, 10):.0f} p90 {np.percentile(tot, 90):.0f}); "
         f"avg resident blocks/CU {conc:.2f}")
    emit(f"    prologue    {pro.mean():8.0f} ns ({100 * pro.mean() / tot.mean():.0f}%)")
    emit(f"    K loop      {loop.mean():8.0f} ns ({100 * loop.mean() / tot.mean():.0f}%) = {loop.mean() / nk:.0f} ns per K-tile: "
         f"mma {mma.mean():.0f} ns + wait/barrier {wait.mean():.0f} ns  [per-tile mma: "
         + " ".join(f"{x:.0f}" for x in mma.mean(0)) + " | wait: " + " ".join(f"{x:.0f}" for x in wait.mean(0)) + "]")
    emit(f"    epilogue    {epi.mean():8.0f} ns ({100 * epi.mean() / tot.mean():.0f}%): operand loads + cvt + LDS staging "
         f"{ns(t[:, 24] - t[:, 20]).mean():.0f}, barrier {ns(t[:, 25] - t[:, 24]).mean():.0f}, (residual +) store loop "
         f"{ns(t[:, 26] - t[:, 25]).mean():.0f}, store drain {ns(t[:, 21] - t[:, 26]).mean():.0f}")


T = 196608
run("attn out-proj (L0)", T, 320, 320, res=True)
run("QKV (L0)", T, 960, 320)
run("GEGLU (L0)", T, 2560, 320, act=3)
run("FF down (L0)", T, 320, 1280, res=True)
run("GEGLU (L2)", 12288, 10240, 1280, act=3)
run("QKV-like long K", T, 960, 2560)
run("conv3x3 320 (L0)", T, 320, 2880, mode=1, conv=(64, 64, 64, 64, 1, 0), res=True)
run("temporal conv (L0)", T, 320, 960, mode=2, temporal=(16, 4096))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_trace.txt"), "w").write("\n".join(lines) + "\n")
