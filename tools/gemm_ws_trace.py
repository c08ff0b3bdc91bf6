"""Phase timeline of the weight-stationary kernel (probe build, debug flag bit5): third strip of waves 0 and 4 of every block --
ticks per K-step, K loop total, epilogue.  gpurun_out/gemm_ws_trace.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from anyv2v_amd import _lib, ops  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "libanyv2v_hip_experiments.so")
dev = "cuda"
lines = []


def run(tag, M, N, act=0, res=False, extra=0):
    K = 320
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device=dev)
    n_out = N // 2 if act == 3 else N
    out = torch.empty(M, n_out, dtype=torch.float16, device=dev)
    r = torch.randn(M, n_out, device=dev).half() if res else None
    kw = dict(bias=b, out=out, act=act, residual=r)
    ws = ops._workspace(torch.device(dev, 0))
    ops.GEMM_FLAGS = 0
    for _ in range(2):
        ops.gemm(a, w, **kw)
    ops.GEMM_FLAGS = 32 | extra
    ws.zero_()
    ops.gemm(a, w, **kw)
    torch.cuda.synchronize()
    ops.GEMM_FLAGS = 0
    t = ws.view(torch.int64)[: 256 * 2 * 16].cpu().numpy().reshape(512, 16)
    t = t[t[:, 0] != 0]
    steps = np.diff(t[:, 0:11], axis=1)
    epi = t[:, 11] - t[:, 10]
    s = (f"{tag}: M={M} N={N}: {len(t)} traced waves; ticks per K-step mean [" + " ".join(f"{x:.0f}" for x in steps.mean(0)) + "] "
         f"K loop {steps.sum(1).mean():.0f} (min {steps.sum(1).min():.0f} max {steps.sum(1).max():.0f}), epilogue {epi.mean():.0f} "
         f"(min {epi.min():.0f} max {epi.max():.0f})")
    lines.append(s)
    print(s, flush=True)


run("B3 proj", 196608, 320)
run("B3 out-proj +res", 196608, 320, res=True)
run("B3 QKV", 196608, 960)
run("B3 GEGLU", 196608, 2560, act=3)
run("B1 QKV", 65536, 960)
run("B3 GEGLU, 4 waves per block (one per SIMD)", 196608, 2560, act=3, extra=8192)
run("B3 GEGLU, 1 wave per block", 196608, 2560, act=3, extra=16384)
run("B3 QKV, 4 waves per block", 196608, 960, extra=8192)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_ws_trace.txt"), "w").write("\n".join(lines) + "\n")
