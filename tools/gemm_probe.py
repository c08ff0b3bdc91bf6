"""GEMM microbench (GPU): times the gather-GEMM variants on the UNet's dominant shapes.  Used standalone and under
`rocprofv3 --pmc ...` (few launches per case so counter runs stay short).  Writes gpurun_out/gemm_probe.txt.

    python tools/gemm_probe.py [--quick] [--variants 0,8,4,r]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
quick = "--quick" in sys.argv
variants = "0,8,4,r"
for i, a in enumerate(sys.argv):
    if a == "--variants":
        variants = sys.argv[i + 1]
lines = []


def timeit(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def case(tag, M, N, K, iters, mode=0, conv=None, act=0, rows_in=None):
    a = torch.randn(rows_in or M, K, device=dev).half()
    taps = 9 if mode == 1 else (3 if mode == 2 else 1)
    w = (torch.randn(N, taps * K, device=dev) / (taps * K) ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device=dev)
    out = torch.empty(M, N // 2 if act == 3 else N, dtype=torch.float16, device=dev)
    fn = lambda: ops.gemm(a, w, bias=b, out=out, mode=mode, conv=conv, M=M, act=act,
                          temporal=(16, M // 48) if mode == 2 else None)
    ms = timeit(fn, 2 if quick else iters, warm=1 if quick else 2)
    tf = 2.0 * M * N * K * taps / (ms * 1e-3) / 1e12
    lines.append(f"{tag:<46s} M={M:6d} N={N:5d} K={K:4d}x{taps}: {ms:8.3f} ms  {tf:7.1f} TFLOP/s")
    print(lines[-1], flush=True)


T, H = 48 * 4096, 64
for v in variants.split(","):
    if v == "r":
        ops.USE_GLDS, ops.GEMM_FLAGS = False, 0
    else:
        ops.USE_GLDS, ops.GEMM_FLAGS = True, int(v)
    tag = f"[{'reg' if v == 'r' else 'glds flags=' + v}] "
    case(tag + "conv3x3 320->320 @64x64", T, 320, 320, 10, mode=1, conv=(H, H, H, H, 1, 0))
    if quick:
        continue
    case(tag + "linear 320->320", T, 320, 320, 20)
    case(tag + "linear 320->960 (qkv)", T, 960, 320, 20)
    case(tag + "linear 1280->320 (ff down)", T, 320, 1280, 10)
    case(tag + "geglu 320->2560", T, 2560, 320, 10, act=3)
    case(tag + "temporal conv 320 @64x64", T, 320, 320, 10, mode=2)
    case(tag + "conv3x3 640->640 @32x32", T // 4, 640, 640, 10, mode=1, conv=(32, 32, 32, 32, 1, 0))
    case(tag + "conv3x3 1280->1280 @16x16", T // 16, 1280, 1280, 10, mode=1, conv=(16, 16, 16, 16, 1, 0))
    case(tag + "conv3x3 1280->1280 @8x8", T // 64, 1280, 1280, 10, mode=1, conv=(8, 8, 8, 8, 1, 0))
    case(tag + "linear 1280->1280 @16x16", T // 16, 1280, 1280, 20)

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_probe_quick.txt" if quick else "gemm_probe.txt"), "w").write("\n".join(lines) + "\n")
