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
# knock-outs (probe build, `make -C anyv2v_amd/csrc experiments`): what each part of a slab step costs
exp = os.path.join(ROOT, "tools", "libanyv2v_hip_experiments.so")
if os.path.isfile(exp) and os.environ.get("FF_KO", "1") == "1":
    import ctypes as C
    from anyv2v_amd import _lib
    prod = _lib._lib
    _lib._lib, _lib.LIB_PATH = None, exp
    lib = _lib.load()
    _lib._lib = prod
    M = 196608
    x, r = gc.rnd(M, C_ := 320), gc.rnd(M, 320)
    y = torch.empty(M, 320, dtype=torch.float16, device=dev)
    d = _lib.FFDesc()
    d.X, d.W1, d.b1, d.W2, d.b2, d.Y, d.R = x.data_ptr(), w1p.data_ptr(), b1p.data_ptr(), w2s.data_ptr(), b2.data_ptr(), y.data_ptr(), r.data_ptr()
    d.M, d.C, d.H, d.ldx, d.ldy, d.ldr = M, 320, 1280, 320, 320, 320
    names = {0: "full", 1: "no LDS-DMA", 2: "no erf-GELU (h * gate)", 3: "no phase-B MFMAs", 4: "no phase-A MFMAs", 5: "no step barrier", 6: "no exchange read",
             8: "VARIANT deeper rings, no setprio", 16: "VARIANT no setprio", 24: "VARIANT deeper rings + setprio"}
    ts = {k: [] for k in names}
    for _ in range(4):
        for ko in names:
            d.flags = ko
            fn = lambda: lib.anyv2v_ff_geglu_f16(C.byref(d), ops._stream())
            fn()
            ts[ko].append(timeit(fn, 5))
    say("knock-outs at M = 196608 (probe build; results of the knocked-out forms are wrong by construction): " +
        " | ".join(f"{names[k]} {sorted(v)[len(v) // 2]:.0f} us" for k, v in ts.items()))
    # phase stamps (s_memtime ticks) of waves 0 and 4 of block 0, steps 4..11 of the second round
    import numpy as np
    d.flags = 7
    lib.anyv2v_ff_geglu_f16(C.byref(d), ops._stream())
    torch.cuda.synchronize()
    C.CDLL(exp)  # same handle
    rd_ = getattr(lib, "anyv2v_ff_trace_read")
    rd_.restype, rd_.argtypes = C.c_int, [C.c_void_p]
    buf = np.zeros(2 * 8 * 16, dtype=np.int64)
    rd_(buf.ctypes.data)
    tr = buf.reshape(2, 8, 16)
    say("stamps: 0 step start, 1 DMA issued, 2 phase-A MFMAs done, 3 GEGLU + exchange write done, 4 phase-B start, 5 phase-B done, 6 before the wait, 7 after vmcnt/lgkmcnt(0), 8 after the barrier")
    for wv, nm in ((0, "wave 0 (A, B, barrier)"), (1, "wave 4 (A, barrier, B)")):
        for st in range(8):
            t = tr[wv, st]
            base = t[0]
            say(f"  {nm} step {st + 4}: " + " ".join(f"{k}:{int(t[k] - base):5d}" for k in range(9)))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "ff_fused_ab.txt"), "w").write("\n".join(lines) + "\n")
