"""A/B (GPU) of the two split-K paths on the 8x8-level shapes (N = 1280, M = 3072 / 1024): 128-row kernel + split-K
(flags 4) vs split-K work items of the persistent 192x320 kernel (flags 0).  Writes gpurun_out/gemm_split_ab.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []


def timeit(fn, iters=20, warm=3):
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


def case(tag, M, N, K, mode=0, conv=None, temporal=None, res=False):
    taps = {0: 1, 1: 9, 2: 3}[mode]
    a = torch.randn(M, K // taps, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device=dev)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    r = torch.randn(M, N, device=dev).half() if res else None
    kw = dict(bias=b, out=out, mode=mode, conv=conv, temporal=temporal, residual=r, M=M)
    us, ys = [], []
    for flags in (4, 0, 16):
        ops.GEMM_FLAGS = flags
        us.append(timeit(lambda: ops.gemm(a, w, **kw)))
        ys.append(out.float().clone())
    fl = 2.0 * M * N * K
    diff = (ys[0] - ys[1]).abs().max().item()
    lines.append(f"{tag:<28s} M={M:5d} N={N:5d} K={K:5d}: 128-row split {us[0]:7.1f} us ({fl / us[0] / 1e6:5.0f} TF) | persistent split "
                 f"{us[1]:7.1f} us ({fl / us[1] / 1e6:5.0f} TF) | no split {us[2]:7.1f} us | max |diff| {diff:.2e}")
    print(lines[-1], flush=True)


for B in (3, 1):
    M = B * 1024
    case(f"B{B} 8x8 conv3x3 1280 +res", M, 1280, 11520, mode=1, conv=(8, 8, 8, 8, 1, 0), res=True)
    case(f"B{B} 8x8 conv3x3 2560 (cat)", M, 1280, 23040, mode=1, conv=(8, 8, 8, 8, 1, 0))
    case(f"B{B} 8x8 temporal conv", M, 1280, 3840, mode=2, temporal=(16, 64))
    case(f"B{B} 8x8 FF down +res", M, 1280, 5120, res=True)
    case(f"B{B} 8x8 out-proj +res", M, 1280, 1280, res=True)
    case(f"B{B} 8x8 shortcut 2560", M, 1280, 2560)
ops.GEMM_FLAGS = 0
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_split_ab.txt"), "w").write("\n".join(lines) + "\n")
