"""Interleaved A/B of the one-wave-per-SIMD persistent kernel (gemm_sw.hip, flags bit21) against the default dispatch on the step
pair's GEMM shapes (B = 3 edit step and B = 1 inversion step), random fp16 operands.  Writes gpurun_out/r06_gemm_sw_ab.txt.
    python tools/gemm_sw_ab.py [--rounds 6] [--tag name]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--tag", default="r06_gemm_sw_ab")
ap.add_argument("--quick", action="store_true")
ap.add_argument("--sk", action="store_true", help="second arm = the stream-K form (flags bit27) instead of the whole-tile form (bit21)")
ap.add_argument("--batches", default="3,1")
args = ap.parse_args()
dev = "cuda"
lines = []
ARMS = (("default", 0), ("sw", 1 << 27 if args.sk else 1 << 21))


def case(tag, M, N, K, mode=0, act=0, conv=None, temporal=None, res=False, rv=0, a_rows=None, c1=0):
    taps = {0: 1, 1: 9, 2: 3}[mode]
    cin = K // taps
    a = torch.randn(a_rows or M, cin - c1, device=dev).half()
    a1 = torch.randn(a_rows or M, c1, device=dev).half() if c1 else None
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.randn(N, device=dev).half()
    n_out = N // 2 if act == 3 else N
    outs = [torch.empty(M, n_out, dtype=torch.float16, device=dev) for _ in ARMS]
    r = torch.randn(M, n_out, device=dev).half() if res else None
    rowvec = torch.randn(M // rv, N, device=dev).half() if rv else None
    kw = dict(bias=b, mode=mode, act=act, conv=conv, temporal=temporal, residual=r, M=M, rowvec=rowvec, rowvec_div=rv, a1=a1)
    times = [[] for _ in ARMS]
    for i, (_, flags) in enumerate(ARMS):
        ops.GEMM_FLAGS = flags
        for _ in range(2):
            ops.gemm(a, w, out=outs[i], **kw)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for i, (_, flags) in enumerate(ARMS):
            ops.GEMM_FLAGS = flags
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm(a, w, out=outs[i], **kw)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 4 * 1e3)
    ops.GEMM_FLAGS = 0
    med = [sorted(t)[len(t) // 2] for t in times]
    eq = bool(torch.equal(outs[0], outs[1]))
    err = float((outs[0].float() - outs[1].float()).abs().max() / outs[0].float().abs().max())
    fl = 2.0 * M * N * K
    lines.append(f"{tag:<30s} M={M:6d} N={N:5d} K={K:5d}: default {med[0]:7.1f} us ({fl / med[0] / 1e6:6.0f} TF) | sw {med[1]:7.1f} us "
                 f"({fl / med[1] / 1e6:6.0f} TF) | sw/default {med[1] / med[0]:5.3f} | bit-equal {eq} (max diff {err:.1e})")
    print(lines[-1], flush=True)


for B, tagB in [(int(b), f"B{b}") for b in args.batches.split(",")]:
    T0, T1, T2, T3 = B * 65536, B * 16384, B * 4096, B * 1024
    if not args.quick:
        case(f"{tagB} L0 conv3x3 +res", T0, 320, 2880, mode=1, conv=(64, 64, 64, 64, 1, 0), res=True)
        case(f"{tagB} L0 conv3x3 +temb", T0, 320, 2880, mode=1, conv=(64, 64, 64, 64, 1, 0), rv=65536)
        case(f"{tagB} L0 conv3x3 640->320 +temb", T0, 320, 5760, mode=1, conv=(64, 64, 64, 64, 1, 0), rv=65536, c1=320)
        case(f"{tagB} L0 temporal conv", T0, 320, 960, mode=2, temporal=(16, 4096))
        case(f"{tagB} L0 FF down +res", T0, 320, 1280, res=True)
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
    if not args.quick:
        case(f"{tagB} L3 out-proj +res", T3, 1280, 1280, res=True)
        case(f"{tagB} L3 QKV", T3, 3840, 1280)
        case(f"{tagB} L3 GEGLU", T3, 10240, 1280, act=3)
        case(f"{tagB} L3 FF down +res", T3, 1280, 5120, res=True)
        case(f"{tagB} L3 conv3x3 +res", T3, 1280, 11520, mode=1, conv=(8, 8, 8, 8, 1, 0), res=True)
        case(f"{tagB} L3 conv3x3 2560->1280 +temb", T3, 1280, 23040, mode=1, conv=(8, 8, 8, 8, 1, 0), rv=64, c1=1280)
        case(f"{tagB} L3 temporal conv", T3, 1280, 3840, mode=2, temporal=(16, 64))
        case(f"{tagB} L2 conv3x3 2560->1280 +temb", T2, 1280, 23040, mode=1, conv=(16, 16, 16, 16, 1, 0), rv=256, c1=1280)
        case(f"{tagB} L1 conv3x3 1280->640 +temb", T1, 640, 11520, mode=1, conv=(32, 32, 32, 32, 1, 0), rv=1024, c1=640)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", args.tag + ".txt"), "w").write("\n".join(lines) + "\n")
