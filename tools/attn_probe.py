"""Attention microbench (GPU): spatial self-attention kernel variants at the UNet's shapes.
Writes gpurun_out/attn_probe.txt.   python tools/attn_probe.py [--quick]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
quick = "--quick" in sys.argv
lines = []


def timeit(fn, iters, warm=5):
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


def attn_case(tag, N, h, S, iters, Sk=None, kv_div=1, qk_mod=0):
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
    ms = timeit(fn, iters)
    tf = 4.0 * N * h * S * Sk * 64 / (ms * 1e-3) / 1e12
    lines.append(f"{tag:<40s} N={N:3d} h={h:2d} S={S:5d} Sk={Sk:5d}: {ms:8.3f} ms  {tf:7.1f} TFLOP/s")
    print(lines[-1], flush=True)


# the first case of a process otherwise measures the clock ramp (seen: +15 %): spin the GPU up first
_w = torch.randn(48 * 4096, 960, device=dev).half()
_o = torch.empty(48 * 4096, 320, dtype=torch.float16, device=dev)
for _ in range(40):
    ops.attention(_w[:, :320], _w[:, 320:640], _w[:, 640:], _o, batch=48, heads=5, Sq=4096, Sk=4096, inner=1,
                  q_strides=(4096, 0, 1), kv_strides=(4096, 0, 1))
torch.cuda.synchronize()
del _w, _o
for flags, name in ((0, "v2"),) + tuple((int(f), f"flag {f}") for f in os.environ.get("ATTN_PROBE_FLAGS", "").split(",") if f):
    ops.ATTN_FLAGS = flags
    attn_case(f"[{name}] spatial 64x64 B=3", 48, 5, 4096, 10)
    attn_case(f"[{name}] spatial 64x64 B=1", 16, 5, 4096, 10)
    attn_case(f"[{name}] spatial 32x32 B=3", 48, 10, 1024, 20)
    attn_case(f"[{name}] spatial 16x16 B=3", 48, 20, 256, 20)
    attn_case(f"[{name}] cross 64x64 Sk=145 B=3", 48, 5, 4096, 20, Sk=145, kv_div=16)
for flags, name in ((0, "shared softmax"), (8, "v2 per-branch aliasing")):
    ops.ATTN_FLAGS = flags
    attn_case(f"[{name}] spatial 64x64 B=3 PnP inject", 48, 5, 4096, 10, qk_mod=16)
    attn_case(f"[{name}] spatial 32x32 B=3 PnP inject", 48, 10, 1024, 20, qk_mod=16)
ops.ATTN_FLAGS = 0

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "attn_probe.txt"), "w").write("\n".join(lines) + "\n")
