"""Phase timeline of the persistent 192x320 GEMM kernel (debug flag bit5): first tile of every block.
Per K-tile: wait (vmcnt + barrier at the top) and MFMA-stream time; then the epilogue.  gpurun_out/gemm_big_trace.txt"""
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


def run(tag, M, N, K, mode=0, act=0, conv=None, temporal=None, res=False):
    taps = {0: 1, 1: 9, 2: 3}[mode]
    a = torch.randn(M, K // taps, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device=dev)
    n_out = N // 2 if act == 3 else N
    out = torch.empty(M, n_out, dtype=torch.float16, device=dev)
    r = torch.randn(M, n_out, device=dev).half() if res else None
    kw = dict(bias=b, out=out, mode=mode, act=act, conv=conv, temporal=temporal, residual=r, M=M)
    ws = ops._workspace(torch.device(dev, 0))
    ops.GEMM_FLAGS = 8
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
    ops.GEMM_FLAGS = 8 | 32
    ws.zero_()
    ops.gemm(a, w, **kw)
    torch.cuda.synchronize()
    ops.GEMM_FLAGS = 0
    nblk = min(256, ((M + 191) // 192) * (N // 320))
    t = ws.view(torch.int64)[: nblk * 32].cpu().numpy().reshape(nblk, 32)
    nk = int(t[0, 28])
    if nk == 0:  # (the traced tile index does not exist for this launch: AV_TRACE_TILE rounds per block needed)
        print(f'{tag}: no traced tile', flush=True)
        return
    kk = min(nk, 8)
    wait = np.stack([t[:, 3 + 3 * k] - t[:, 2 + 3 * k] for k in range(kk)], 1)
    mma = np.stack([t[:, 4 + 3 * k] - t[:, 3 + 3 * k] for k in range(kk)], 1)
    epi = t[:, 27] - t[:, 26]
    settle, bar, body = (t[:, 30] - t[:, 26]).mean(), (t[:, 31] - t[:, 30]).mean(), (t[:, 27] - t[:, 31]).mean()
    vm3 = (t[:, 29] - t[:, 2 + 9]).mean() if nk > 3 else -1
    tiles = ((M + 191) // 192) * (N // 320)
    s = (f"{tag}: M={M} N={N} K={K}: {us:.1f} us ({2.0 * M * N * K / us / 1e6:.0f} TF/s), {tiles} tiles / {nblk} blocks, {nk} K-tiles; "
         f"per K-tile (ticks): wait+barrier [" + " ".join(f"{x:.0f}" for x in wait.mean(0)) + "]  MFMA stream [" +
         " ".join(f"{x:.0f}" for x in mma.mean(0)) + f"]  (MFMA floor 2 waves/SIMD: {120 * 12.4:.0f}); of K-tile 3's wait, vmcnt(0) took {vm3:.0f} (wave 0); epilogue {epi.mean():.0f} ticks = prefetch settle {settle:.0f} + barrier {bar:.0f} + convert/turn/store {body:.0f}")
    lines.append(s)
    print(s, flush=True)


T = 196608
run("L0 GEGLU", T, 2560, 320, act=3)
run("L0 QKV", T, 960, 320)
run("L0 out-proj +res", T, 320, 320, res=True)
run("L0 conv3x3 +res", T, 320, 2880, mode=1, conv=(64, 64, 64, 64, 1, 0), res=True)
run("L2 conv3x3 +res", 12288, 1280, 11520, mode=1, conv=(16, 16, 16, 16, 1, 0), res=True)
run("L2 GEGLU", 12288, 10240, 1280, act=3)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_big_trace.txt"), "w").write("\n".join(lines) + "\n")
