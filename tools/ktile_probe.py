"""Per-block fixed cost vs per-K-tile cost of the GEMM kernel: time(M, N, K) for K = 64..640 at fixed M, N."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from anyv2v_amd import ops
dev = "cuda"
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
T = 196608
lines = []
for flags in (4, 0):
    ops.GEMM_FLAGS = flags
    for N in (320, 960, 2560):
        for K in (64, 128, 192, 320, 640):
            a = torch.randn(T, K, device=dev).half(); w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
            b = torch.zeros(N, dtype=torch.float16, device=dev); out = torch.empty(T, N, dtype=torch.float16, device=dev)
            us = timeit(lambda: ops.gemm(a, w, bias=b, out=out))
            mb = (T * K + T * N) * 2 / 1e6
            lines.append(f"flags={flags} N={N:5d} K={K:4d}: {us:8.1f} us   {mb/us*1e-3*1e3:7.1f} GB/s... {mb:7.1f} MB  {2*T*N*K/us/1e6:7.1f} TF")
            print(lines[-1], flush=True)
open(os.path.join(ROOT, "gpurun_out", "ktile_probe.txt"), "w").write("\n".join(lines))
