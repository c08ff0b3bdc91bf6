"""How far are the hand-written GEMM kernels from the vendor library?  torch.matmul (hipBLASLt / rocBLAS, fp16 in,
fp32 accumulate) vs ops.gemm on the workload's large plain-linear shapes.  gpurun_out/vendor_gemm_probe.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, N, K) in [(196608, 320, 320), (196608, 960, 320), (196608, 2560, 320), (196608, 320, 1280), (196608, 320, 2880),
                  (49152, 1920, 640), (49152, 640, 2560), (49152, 640, 5760), (12288, 3840, 1280), (12288, 1280, 5120),
                  (12288, 1280, 11520), (65536, 960, 320), (4096, 1280, 1280), (16384, 640, 640)]:
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    wt = w.t().contiguous()
    t_mine = timeit(lambda: ops.gemm(a, w, out=out))
    t_nt = timeit(lambda: torch.matmul(a, w.t(), out=out))
    t_nn = timeit(lambda: torch.matmul(a, wt, out=out))
    fl = 2.0 * M * N * K
    lines.append(f"M={M:6d} N={N:5d} K={K:5d}: ours {t_mine:7.1f} us ({fl / t_mine / 1e6:6.0f} TF) | torch A@W^T {t_nt:7.1f} us "
                 f"({fl / t_nt / 1e6:6.0f} TF) | torch A@Wt {t_nn:7.1f} us ({fl / t_nn / 1e6:6.0f} TF)")
    print(lines[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "vendor_gemm_probe.txt"), "w").write("\n".join(lines) + "\n")
