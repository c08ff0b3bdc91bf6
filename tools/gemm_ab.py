"""A/B of the GEMM kernels (GPU) on the bench workload's large shapes: 128-row kernel (flags 4) vs persistent 256x320
kernel (flags 0 = dispatch heuristic, flags 8 = forced).  Writes gpurun_out/gemm_ab.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []


def timeit(fn, iters=10, warm=2):
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


def case(tag, M, N, K, mode=0, act=0, conv=None, temporal=None, res=False, rv=0, a_rows=None):
    taps = {0: 1, 1: 9, 2: 3}[mode]
    a = torch.randn(a_rows or M, K // taps, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device=dev)
    n_out = N // 2 if act == 3 else N
    out = torch.empty(M, n_out, dtype=torch.float16, device=dev)
    r = torch.randn(M, n_out, device=dev).half() if res else None
    rowvec = torch.randn(M // rv, N, device=dev).half() if rv else None
    kw = dict(bias=b, out=out, mode=mode, act=act, conv=conv, temporal=temporal, residual=r, M=M, rowvec=rowvec, rowvec_div=rv)
    res_us = []
    for flags in (4, 8, 0):
        ops.GEMM_FLAGS = flags
        res_us.append(timeit(lambda: ops.gemm(a, w, **kw)))
    fl = 2.0 * M * N * K
    lines.append(f"{tag:<34s} M={M:6d} N={N:5d} K={K:5d}: old {res_us[0]:7.1f} us ({fl / res_us[0] / 1e6:6.0f} TF) | big-frc {res_us[1]:7.1f} us "
                 f"({fl / res_us[1] / 1e6:6.0f} TF) | big-auto {res_us[2]:7.1f} us ({fl / res_us[2] / 1e6:6.0f} TF)")
    print(lines[-1], flush=True)


for B, tagB in ((3, "B3"), (1, "B1")):
    T0, T1, T2 = B * 65536, B * 16384, B * 4096
    case(f"{tagB} L0 out-proj +res", T0, 320, 320, res=True)
    case(f"{tagB} L0 QKV", T0, 960, 320)
    case(f"{tagB} L0 GEGLU", T0, 2560, 320, act=3)
    case(f"{tagB} L0 FF down +res", T0, 320, 1280, res=True)
    case(f"{tagB} L0 conv3x3 +res", T0, 320, 2880, mode=1, conv=(64, 64, 64, 64, 1, 0), res=True)
    case(f"{tagB} L0 conv3x3 +temb", T0, 320, 2880, mode=1, conv=(64, 64, 64, 64, 1, 0), rv=65536)
    case(f"{tagB} L0 temporal conv", T0, 320, 960, mode=2, temporal=(16, 4096))
    case(f"{tagB} L1 out-proj +res", T1, 640, 640, res=True)
    case(f"{tagB} L1 QKV", T1, 1920, 640)
    case(f"{tagB} L1 GEGLU", T1, 5120, 640, act=3)
    case(f"{tagB} L1 FF down +res", T1, 640, 2560, res=True)
    case(f"{tagB} L1 conv3x3 +res", T1, 640, 5760, mode=1, conv=(32, 32, 32, 32, 1, 0), res=True)
    case(f"{tagB} L1 temporal conv", T1, 640, 1920, mode=2, temporal=(16, 1024))
    case(f"{tagB} L2 out-proj +res", T2, 1280, 1280, res=True)
    case(f"{tagB} L2 QKV", T2, 3840, 1280)
    case(f"{tagB} L2 GEGLU", T2, 10240, 1280, act=3)
    case(f"{tagB} L2 FF down +res", T2, 1280, 5120, res=True)
    case(f"{tagB} L2 conv3x3 +res", T2, 1280, 11520, mode=1, conv=(16, 16, 16, 16, 1, 0), res=True)
    case(f"{tagB} L2 temporal conv", T2, 1280, 3840, mode=2, temporal=(16, 256))
ops.GEMM_FLAGS = 0
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_ab.txt"), "w").write("\n".join(lines) + "\n")
