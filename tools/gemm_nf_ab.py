"""A/B of the 128-row GEMM kernel's tile width (160 vs 128 columns, flags bit12 / bit11) on the small-M launches of the step
(16x16 and 8x8 levels, B = 1 and B = 3): interleaved rounds in one process.  Writes gpurun_out/gemm_nf_ab.txt.
    python tools/gemm_nf_ab.py"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(tag, M, N, K, mode=0, res=False, temporal=None, conv=None):
    a = torch.randn(M, K // (3 if mode == 2 else 9 if mode == 1 else 1), device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device=dev)
    r = torch.randn(M, N, device=dev).half() if res else None
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    kw = dict(bias=b, residual=r, out=out, mode=mode)
    if mode == 2:
        kw["temporal"] = temporal
    if mode == 1:
        kw["conv"] = conv
    res_t = {}
    outs = {}
    for rnd in range(5):
        for name, fl in (("nf5", 4096), ("nf4", 2048), ("auto", 0)):
            ops.GEMM_FLAGS = fl
            res_t.setdefault(name, []).append(timeit(lambda: ops.gemm(a, w, **kw)))
            outs[name] = out.clone()
    ops.GEMM_FLAGS = 0
    same = torch.equal(outs["nf5"], outs["nf4"]) and torch.equal(outs["nf5"], outs["auto"])
    m = {k: statistics.median(v) for k, v in res_t.items()}
    line = (f"{tag:<34s} M={M:6d} N={N:5d} K={K:6d}: nf5 {m['nf5']:7.1f} us | nf4 {m['nf4']:7.1f} us | auto {m['auto']:7.1f} us"
            f" | nf4/nf5 x{m['nf5'] / m['nf4']:.2f} | bit-equal {same}")
    print(line, flush=True)
    lines.append(line)


for B, tagB in ((1, "B1"), (3, "B3")):
    M16, M8, M32 = B * 16 * 256, B * 16 * 64, B * 16 * 1024
    case(f"{tagB} 16x16 QKV", M16, 3840, 1280)
    case(f"{tagB} 16x16 proj +res", M16, 1280, 1280, res=True)
    case(f"{tagB} 16x16 FF-down +res", M16, 1280, 5120, res=True)
    case(f"{tagB} 16x16 temporal conv", M16, 1280, 3840, mode=2, temporal=(16, 256))
    case(f"{tagB} 16x16 conv3x3 +res", M16, 1280, 11520, mode=1, res=True, conv=(16, 16, 16, 16, 1, 0))
    case(f"{tagB} 8x8 proj +res", M8, 1280, 1280, res=True)
    case(f"{tagB} 8x8 temporal conv", M8, 1280, 3840, mode=2, temporal=(16, 64))
    case(f"{tagB} 8x8 conv3x3 +res", M8, 1280, 11520, mode=1, res=True, conv=(8, 8, 8, 8, 1, 0))
    case(f"{tagB} 32x32 QKV", M32, 1920, 640)
    case(f"{tagB} 32x32 proj +res", M32, 640, 640, res=True)
    case(f"{tagB} 32x32 FF-down +res", M32, 640, 2560, res=True)
    case(f"{tagB} 32x32 temporal conv", M32, 640, 1920, mode=2, temporal=(16, 1024))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_nf_ab.txt"), "w").write("\n".join(lines) + "\n")
