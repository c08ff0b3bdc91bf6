"""A/B of forms of the GEMM kernels selected by AnyV2VGemmDesc.flags and / or by build (GEMM_FORMS="flag:name[:path/to/lib.so],...";
a third field loads another build of the library with the same ABI next to the product one, in the same process): parity of each
on the big-tile check, then interleaved timing (rounds in one process) on the bench workload's large shapes.
Writes gpurun_out/gemm_epi_ab.txt.   python tools/gemm_epi_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import gpu_checks as gc  # noqa: E402
from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []
from anyv2v_amd import _lib  # noqa: E402

_product = _lib.load()
_handles = {}
FORMS = []
for spec in os.environ.get("GEMM_FORMS", "0:product").split(","):
    f = spec.split(":")
    if len(f) > 2:  # another build
        _lib._lib, _lib.LIB_PATH = None, os.path.join(ROOT, f[2])
        _handles[f[1]] = _lib.load()
        _lib._lib = _product
    else:
        _handles[f[1]] = _product
    FORMS.append((int(f[0]), f[1]))


def select(flag, name):
    ops.GEMM_FLAGS = flag
    _lib._lib = _handles[name]


def say(s):
    lines.append(s)
    print(s, flush=True)


for flag, name in ([] if os.environ.get("GEMM_AB_NO_PARITY", "0") == "1" else FORMS):   # (parity of every form; skipped when a test run covers it)
    select(flag, name)
    res = gc.check_gemm_big() + gc.check_conv(("glds",)) + gc.check_gemm_splitk() + gc.check_vae_kernels()
    bad = [r for r in res if not r["ok"]]
    say(f"[{name}] parity: {len(res) - len(bad)}/{len(res)} ok, worst {max(r['err'] for r in res):.2e}")
    for r in bad:
        say(f"    FAIL {r['name']}: {r['err']:.3e} > {r['tol']:.1e}")


def timeit(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(tag, M, N, K, mode=0, act=0, conv=None, temporal=None, res=False, rv=0, rounds=5, iters=10, a_rows=None):
    taps = {0: 1, 1: 9, 2: 3}[mode]
    a = torch.randn(a_rows or M, K // taps, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device=dev)
    n_out = N // 2 if act == 3 else N
    out = torch.empty(M, n_out, dtype=torch.float16, device=dev)
    r = torch.randn(M, n_out, device=dev).half() if res else None
    rowvec = torch.randn(M // rv, N, device=dev).half() if rv else None
    kw = dict(bias=b, out=out, mode=mode, act=act, conv=conv, temporal=temporal, residual=r, M=M, rowvec=rowvec, rowvec_div=rv)
    fn = lambda: ops.gemm(a, w, **kw)
    for _ in range(3):
        fn()
    ts = {n: [] for _, n in FORMS}
    for _ in range(rounds):
        for flag, name in FORMS:
            select(flag, name)
            ts[name].append(timeit(fn, iters))
    fl = 2.0 * M * N * K
    by = (M * (K // taps) + N * K + M * n_out * (2 if res else 1)) * 2.0
    say(f"{tag:<26s} M={M:6d} N={N:5d} K={K:5d}: " + " | ".join(
        f"{n}: med {sorted(v)[len(v) // 2]:7.1f} min {min(v):7.1f} us ({fl / min(v) / 1e6:5.0f} TF, {by / min(v) / 1e3:5.0f} GB/s)" for n, v in ts.items()))


if os.environ.get("GEMM_CASES") == "conv":
    for B, tagB in ((3, "B3"), (1, "B1")):
        N0 = 16 * B
        case(f"{tagB} conv 320->320 @64", N0 * 4096, 320, 2880, mode=1, conv=(64, 64, 64, 64, 1, 0), res=True)
        case(f"{tagB} conv 320->320 @64 +temb", N0 * 4096, 320, 2880, mode=1, conv=(64, 64, 64, 64, 1, 0), rv=65536)
        case(f"{tagB} conv 640->320 @64", N0 * 4096, 320, 5760, mode=1, conv=(64, 64, 64, 64, 1, 0), rv=65536)
        case(f"{tagB} conv 960->320 @64", N0 * 4096, 320, 8640, mode=1, conv=(64, 64, 64, 64, 1, 0), rv=65536)
        case(f"{tagB} conv 320->320 s2 64->32", N0 * 1024, 320, 2880, mode=1, conv=(64, 64, 32, 32, 2, 0), a_rows=N0 * 4096)
        case(f"{tagB} conv 640->640 @32", N0 * 1024, 640, 5760, mode=1, conv=(32, 32, 32, 32, 1, 0), res=True)
        case(f"{tagB} conv 1280->640 @32", N0 * 1024, 640, 11520, mode=1, conv=(32, 32, 32, 32, 1, 0), rv=16384)
        case(f"{tagB} conv 1280->1280 @16", N0 * 256, 1280, 11520, mode=1, conv=(16, 16, 16, 16, 1, 0), res=True)
        case(f"{tagB} conv 2560->1280 @16", N0 * 256, 1280, 23040, mode=1, conv=(16, 16, 16, 16, 1, 0), rv=4096)
        case(f"{tagB} conv 1280->1280 @8", N0 * 64, 1280, 11520, mode=1, conv=(8, 8, 8, 8, 1, 0), res=True)
        case(f"{tagB} conv 2560->1280 @8", N0 * 64, 1280, 23040, mode=1, conv=(8, 8, 8, 8, 1, 0), rv=1024)
    select(*FORMS[0])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", os.environ.get("GEMM_AB_OUT", "gemm_epi_ab.txt")), "w").write("\n".join(lines) + "\n")
    sys.exit(0)
for B, tagB in ((3, "B3"), (1, "B1")):
    T0, T1, T2 = B * 65536, B * 16384, B * 4096
    case(f"{tagB} L0 out-proj +res", T0, 320, 320, res=True)
    case(f"{tagB} L0 proj plain", T0, 320, 320)
    case(f"{tagB} L0 QKV", T0, 960, 320)
    case(f"{tagB} L0 GEGLU", T0, 2560, 320, act=3)
    case(f"{tagB} L0 FF down +res", T0, 320, 1280, res=True)
    case(f"{tagB} L0 conv3x3 +res", T0, 320, 2880, mode=1, conv=(64, 64, 64, 64, 1, 0), res=True)
    case(f"{tagB} L0 conv3x3 +temb", T0, 320, 2880, mode=1, conv=(64, 64, 64, 64, 1, 0), rv=65536)
    case(f"{tagB} L0 temporal conv", T0, 320, 960, mode=2, temporal=(16, 4096))
    case(f"{tagB} L0 temporal conv +res", T0, 320, 960, mode=2, temporal=(16, 4096), res=True)
    case(f"{tagB} L1 out-proj +res", T1, 640, 640, res=True)
    case(f"{tagB} L1 QKV", T1, 1920, 640)
    case(f"{tagB} L1 GEGLU", T1, 5120, 640, act=3)
    case(f"{tagB} L1 FF down +res", T1, 640, 2560, res=True)
    case(f"{tagB} L1 conv3x3 +res", T1, 640, 5760, mode=1, conv=(32, 32, 32, 32, 1, 0), res=True)
    case(f"{tagB} L1 temporal conv", T1, 640, 1920, mode=2, temporal=(16, 1024))
    if B == 3:
        case(f"{tagB} L2 QKV", T2, 3840, 1280)
        case(f"{tagB} L2 GEGLU", T2, 10240, 1280, act=3)
        case(f"{tagB} L2 FF down +res", T2, 1280, 5120, res=True)
        case(f"{tagB} L2 conv3x3 +res", T2, 1280, 11520, mode=1, conv=(16, 16, 16, 16, 1, 0), res=True)
select(*FORMS[0])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", os.environ.get("GEMM_AB_OUT", "gemm_epi_ab.txt")), "w").write("\n".join(lines) + "\n")
