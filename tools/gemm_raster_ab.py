"""Tile order of the persistent GEMM kernel on the wide-N launches (GEGLU up-projections at 640 / 1280 channels): classic
(N-fastest, XCD-contiguous) vs rastered super-tiles (AnyV2VGemmDesc.flags bits 13-16), interleaved rounds in one process.
With RASTER_PMC=1 it only launches every form three times (for a `rocprofv3 --pmc FETCH_SIZE` pass; read per dispatch with
tools/pmc_per_dispatch.py gemm_big).   python tools/gemm_raster_ab.py  -> gpurun_out/gemm_raster_ab.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
FORMS = [(1 << 13, "classic"), (3 << 13, "8x4 M-fast"), ((3 << 13) | (1 << 16), "8x4 N-fast"), (2 << 13, "4x8 M-fast"),
         (4 << 13, "16x2 M-fast"), ((4 << 13) | (1 << 16), "16x2 N-fast"), (0, "auto")]
PMC = os.environ.get("RASTER_PMC", "0") == "1"
lines = []


def say(s):
    lines.append(s)
    print(s, flush=True)


def timeit(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(tag, M, N, K, act, rounds=5, iters=10):
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device=dev)
    n_out = N // 2 if act == 3 else N
    out = torch.empty(M, n_out, dtype=torch.float16, device=dev)
    fn = lambda: ops.gemm(a, w, bias=b, out=out, act=act, M=M)
    if PMC:
        for flag, name in FORMS[:-1]:
            ops.GEMM_FLAGS = flag
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        say(f"{tag}: forms {[n for _, n in FORMS[:-1]]} x 3 launches each, in this order")
        return
    ref = None
    ts = {n: [] for _, n in FORMS}
    for flag, name in FORMS:
        ops.GEMM_FLAGS = flag
        fn()
        if ref is None:
            ref = out.clone()
        elif not torch.equal(ref, out):
            say(f"    {tag} [{name}] NOT bit-equal to classic: {float((ref.float() - out.float()).abs().max()):.3e}")
    for _ in range(rounds):
        for flag, name in FORMS:
            ops.GEMM_FLAGS = flag
            ts[name].append(timeit(fn, iters))
    fl = 2.0 * M * N * K
    alg = (M * K + N * K + M * n_out) * 2.0
    say(f"{tag:<22s} M={M:6d} N={N:5d} K={K:5d} alg {alg / 1e6:6.0f} MB: " + " | ".join(
        f"{n}: {sorted(v)[len(v) // 2]:6.1f} us ({fl / sorted(v)[len(v) // 2] / 1e6:4.0f} TF)" for n, v in ts.items()))


for B, tagB in ((3, "B3"), (1, "B1")):
    T1, T2, T3 = B * 16384, B * 4096, B * 1024
    case(f"{tagB} L1 GEGLU", T1, 5120, 640, 3)
    case(f"{tagB} L2 GEGLU", T2, 10240, 1280, 3)
    if not PMC:
        case(f"{tagB} L1 QKV", T1, 1920, 640, 0)
        case(f"{tagB} L2 QKV", T2, 3840, 1280, 0)
ops.GEMM_FLAGS = 0
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_raster_pmc_order.txt" if PMC else "gemm_raster_ab.txt"), "w").write("\n".join(lines) + "\n")
