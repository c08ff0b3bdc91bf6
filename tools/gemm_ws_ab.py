"""A/B of the weight-stationary K = 320 kernel (gemm_ws.hip) against the tile kernels it replaces (flag bit9), on the
bench workload's own launches at the 64x64 level: interleaved rounds in one process, median / min per arm.
Writes gpurun_out/gemm_ws_ab.txt.

    python tools/gemm_ws_ab.py [--rounds 7]
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 7
lines = []
VARIANTS = tuple(int(v) for v in os.environ.get("WS_VARIANTS", "").split(",") if v)   # probe: flags bits 11-12


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(tag, M, N, act=0, res=False, ldc=None, K=320):
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = (torch.randn(N, device=dev) * 0.1).half()
    n_out = N // 2 if act == 3 else N
    buf = torch.empty(M, ldc or n_out, dtype=torch.float16, device=dev)
    out = buf[:, (ldc or n_out) - n_out:]
    r = torch.randn(M, n_out, device=dev).half() if res else None
    kw = dict(bias=b, out=out, act=act, residual=r)
    arms = (512, 0) + tuple(v << 11 for v in VARIANTS)
    t = {f: [] for f in arms}
    for _ in range(rounds):
        for flags in arms:
            ops.GEMM_FLAGS = flags
            t[flags].append(timeit(lambda: ops.gemm(a, w, **kw)))
    fl = 2.0 * M * N * K
    nb = 2.0 * (M * K + N * K + M * n_out * (2 if res else 1))
    old, new = statistics.median(t[512]), statistics.median(t[0])
    lines.append(f"{tag:<26s} M={M:6d} N={N:5d}: tile {old:7.1f} us (min {min(t[512]):7.1f}; {fl / old / 1e6:5.0f} TF, {nb / old / 1e3:5.0f} GB/s)"
                 f" | ws {new:7.1f} us (min {min(t[0]):7.1f}; {fl / new / 1e6:5.0f} TF, {nb / new / 1e3:5.0f} GB/s) | x{old / new:4.2f}"
                 + "".join(f" | var{v} {statistics.median(t[v << 11]):7.1f}" for v in VARIANTS))
    print(lines[-1], flush=True)


def case_ln(tag, M, N, act=0, K=320):
    """LayerNorm kernel + GEMM (weight-stationary, unfused) vs the LayerNorm-folded GEMM."""
    x = torch.randn(M, K, device=dev).half()
    gamma, beta = (1 + 0.1 * torch.randn(K, device=dev)).half(), (0.1 * torch.randn(K, device=dev)).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = (torch.randn(N, device=dev) * 0.1).half()
    wq, bq, c1 = ops.ln_fold(w, b, gamma, beta)
    out = torch.empty(M, N // 2 if act == 3 else N, dtype=torch.float16, device=dev)
    h = torch.empty_like(x)
    ops.GEMM_FLAGS = 0
    t = {"ln+gemm": [], "folded": []}
    for _ in range(rounds):
        t["ln+gemm"].append(timeit(lambda: ops.gemm(ops.layernorm(x, gamma, beta, 1e-5, out=h), w, bias=b, act=act, out=out)))
        t["folded"].append(timeit(lambda: ops.gemm(x, wq, bias=bq, act=act, out=out, ln=(c1, 1e-5))))
    a, f = statistics.median(t["ln+gemm"]), statistics.median(t["folded"])
    lines.append(f"{tag:<26s} M={M:6d} N={N:5d} K={K}: layernorm + gemm {a:7.1f} us | LayerNorm folded {f:7.1f} us | x{a / f:4.2f}")
    print(lines[-1], flush=True)


for B, tagB in ((3, "B3"), (1, "B1")):
    T = B * 65536
    case(f"{tagB} out-proj +res", T, 320, res=True)
    case(f"{tagB} proj (no res)", T, 320)
    case(f"{tagB} QKV", T, 960)
    case(f"{tagB} GEGLU up", T, 2560, act=3)
    if B == 3:
        case(f"{tagB} V-only (2/3 T, ldc 960)", 2 * 65536, 320, ldc=960)
        case(f"{tagB} QKV source third", 65536, 960)
    case_ln(f"{tagB} LN -> QKV", T, 960)
    case_ln(f"{tagB} LN -> to_q", T, 320)
    case_ln(f"{tagB} LN -> GEGLU", T, 2560, act=3)
    case_ln(f"{tagB} LN -> t_in GEGLU", T, 4096, act=3, K=512)
    if B == 3:
        case_ln(f"{tagB} LN -> QKV source third", 65536, 960)
        case_ln(f"{tagB} LN -> V-only 2/3", 131072, 320)
    case(f"{tagB} t_in GEGLU K512", T, 4096, act=3, K=512)
    case(f"{tagB} t_in QKV K512", T, 1536, K=512)
    case(f"{tagB} t_in proj+res K512", T, 512, res=True, K=512)
ops.GEMM_FLAGS = 0
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_ws_ab.txt"), "w").write("\n".join(lines) + "\n")
