"""A/B of flash-kernel variants selected by AnyV2VAttnDesc.flags (SM_FORMS="flag:name,..."; default: 8-wave dispatch vs 4-wave
blocks only) and / or by build (a third field in a form names another .so): parity of each on the full attention check, then interleaved timing at the UNet's shapes.
(Round 2 used it with development flags for the softmax forms: profiles/r02_attn_softmax_forms_ab*.txt.)
Writes gpurun_out/attn_sm_ab.txt.   python tools/attn_sm_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import gpu_checks as gc  # noqa: E402
from anyv2v_amd import _lib, ops  # noqa: E402

dev = "cuda"
lines = []
# SM_FORMS entries: "flag:name" or "flag:name:path/to/other/build.so" (same ABI, loaded next to the product library)
_product = _lib.load()
_handles, FORMS = {}, []
for spec in os.environ.get("SM_FORMS", "0:default,4:4-wave-blocks").split(","):
    f = spec.split(":")
    if len(f) > 2:
        _lib._lib, _lib.LIB_PATH = None, os.path.join(ROOT, f[2])
        _handles[f[1]] = _lib.load()
        _lib._lib = _product
    else:
        _handles[f[1]] = _product
    FORMS.append((int(f[0]), f[1]))


def select(flag, name):
    ops.ATTN_FLAGS = flag
    _lib._lib = _handles[name]



def say(s):
    lines.append(s)
    print(s, flush=True)


for flag, name in FORMS:
    select(flag, name)
    res = gc.check_attention(naive_too=False)
    bad = [r for r in res if not r["ok"]]
    worst = max(res, key=lambda r: r["err"] if r["err"] == r["err"] else 1e9)
    say(f"[{name}] parity: {len(res) - len(bad)}/{len(res)} ok, worst {worst['name']}: {worst['err']:.2e}")
    for r in bad:
        say(f"    FAIL {r['name']}: {r['err']:.3e} > {r['tol']:.1e}")
    for r in res:
        if "forced rescale" in r["name"]:
            say(f"    {r['name']}: {r['err']:.2e}")


def timeit(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def case(tag, N, h, S, iters, qk_mod=0, Sk=None, kv_div=1, rounds=5):
    C = 64 * h
    Sk = Sk or S
    q = torch.randn(N * S, 3 * C, device=dev).half()
    o = torch.empty(N * S, C, dtype=torch.float16, device=dev)
    if Sk == S:
        fn = lambda: ops.attention(q[:, :C], q[:, C:2 * C], q[:, 2 * C:], o, batch=N, heads=h, Sq=S, Sk=S, inner=1,
                                   q_strides=(S, 0, 1), kv_strides=(S, 0, 1), qk_mod=qk_mod)
    else:
        kv = torch.randn((N // kv_div) * Sk, 2 * C, device=dev).half()
        fn = lambda: ops.attention(q[:, :C], kv[:, :C], kv[:, C:], o, batch=N, heads=h, Sq=S, Sk=Sk, inner=1,
                                   q_strides=(S, 0, 1), kv_strides=(Sk, 0, 1), kv_div=kv_div)
    for _ in range(10):
        fn()
    ts = {n: [] for _, n in FORMS}
    for _ in range(rounds):          # interleaved rounds in one process (guide rule 24)
        for flag, name in FORMS:
            select(flag, name)
            ts[name].append(timeit(fn, iters))
    fl = 4.0 * N * h * S * Sk * 64
    say(f"{tag:<34s} " + "  ".join(f"{n}: med {sorted(v)[len(v) // 2]:.3f} min {min(v):.3f} ms ({fl / (min(v) * 1e-3) / 1e12:6.1f} TF)"
                                   for n, v in ts.items()))


case("spatial 64x64 B=3 (graded)", 48, 5, 4096, 10)
case("spatial 64x64 B=1", 16, 5, 4096, 10)
case("spatial 32x32 B=3", 48, 10, 1024, 20)
case("spatial 16x16 B=3", 48, 20, 256, 20)
case("cross 64x64 Sk=145 B=3", 48, 5, 4096, 20, Sk=145, kv_div=16)
case("PnP shared softmax 64x64", 48, 5, 4096, 10, qk_mod=16)
case("PnP shared softmax 32x32", 48, 10, 1024, 20, qk_mod=16)
select(*FORMS[0])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", os.environ.get("SM_AB_OUT", "attn_sm_ab.txt")), "w").write("\n".join(lines) + "\n")
