"""Fused feed-forward kernel (ff_fused.hip) vs the unfused pair (weight-stationary GEGLU GEMM + persistent-kernel Linear) at the
64x64 level's row counts; interleaved rounds in one process.   python tools/ff_fused_ab.py -> gpurun_out/ff_fused_ab.txt"""
import math
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


def say(s):
    lines.append(s)
    print(s, flush=True)


def timeit(fn, iters=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


res = gc.check_ff_fused()
bad = [r for r in res if not r["ok"]]
say(f"parity: {len(res) - len(bad)}/{len(res)} ok, worst {max(r['err'] for r in res):.2e}")
for r in bad:
    say(f"    FAIL {r['name']}: {r['err']:.3e} > {r['tol']:.1e}")
C, H = 320, 1280
w1, b1 = gc.rnd(2 * H, C, scale=1 / math.sqrt(C)), gc.rnd(2 * H, scale=0.1)
w2, b2 = gc.rnd(C, H, scale=1 / math.sqrt(H)), gc.rnd(C, scale=0.1)
w1p, b1p = gc._geglu_pack(w1, b1, H)
w2s = ops.ff_pack_w2(w2)
for M in (196608, 131072, 65536, 32768):
    x, r = gc.rnd(M, C), gc.rnd(M, C)
    y = torch.empty(M, C, dtype=torch.float16, device=dev)
    g = torch.empty(M, H, dtype=torch.float16, device=dev)
    fused = lambda: ops.ff_geglu(x, w1p, b1p, w2s, b2, residual=r, out=y)
    def unfused():
        ops.gemm(x, w1p, bias=b1p, act=ops.ACT_GEGLU, out=g)
        ops.gemm(g, w2, bias=b2, residual=r, out=y)
    for f in (fused, unfused):
        for _ in range(3):
            f()
    tf, tu = [], []
    for _ in range(5):
        tf.append(timeit(fused))
        tu.append(timeit(unfused))
    fl = 2.0 * M * (2 * H * C + H * C)
    med = lambda v: sorted(v)[len(v) // 2]
    say(f"M={M:7d}: fused med {med(tf):7.1f} min {min(tf):7.1f} us ({fl / med(tf) / 1e6:5.0f} TF) | GEGLU GEMM + Linear med {med(tu):7.1f} min {min(tu):7.1f} us "
        f"({fl / med(tu) / 1e6:5.0f} TF)")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "ff_fused_ab.txt"), "w").write("\n".join(lines) + "\n")
