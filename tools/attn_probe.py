"""Microbench probes (GPU): spatial-attention and GEMM kernels under different conditions, to separate kernel
efficiency from clocks / data / cache effects.  Writes gpurun_out/attn_probe.txt."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
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


def attn_case(tag, N, h, S, qkv, iters):
    C = 64 * h
    o = torch.empty(N * S, C, dtype=torch.float16, device=dev)
    fn = lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=N, heads=h, Sq=S, Sk=S, inner=1,
                               q_strides=(S, 0, 1), kv_strides=(S, 0, 1))
    ms = timeit(fn, iters)
    tf = 4.0 * N * h * S * S * 64 / (ms * 1e-3) / 1e12
    lines.append(f"attn {tag:<34s} N={N} h={h} S={S} iters={iters:3d}: {ms:8.3f} ms  {tf:7.1f} TFLOP/s")
    print(lines[-1], flush=True)


def gemm_case(tag, M, N, K, iters, mode=0, conv=None, flags_glds=True, act=0):
    ops.USE_GLDS = flags_glds
    a = torch.randn(M if conv is None else conv[6], K, device=dev).half()
    taps = 9 if mode == 1 else 1
    w = (torch.randn(N, taps * K, device=dev) / (taps * K) ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device=dev)
    out = torch.empty(M, N // 2 if act == 3 else N, dtype=torch.float16, device=dev)
    fn = lambda: ops.gemm(a, w, bias=b, out=out, mode=mode, conv=None if conv is None else conv[:6], M=M, act=act)
    ms = timeit(fn, iters)
    tf = 2.0 * M * N * K * taps / (ms * 1e-3) / 1e12
    lines.append(f"gemm {tag:<34s} M={M} N={N} K={K}x{taps} glds={int(flags_glds)} iters={iters:3d}: {ms:8.3f} ms  {tf:7.1f} TFLOP/s")
    print(lines[-1], flush=True)


N, h, S = 48, 5, 4096
C = 64 * h
if "--gemm-only" in sys.argv:
    N = 1
q_rand = torch.randn(N * S, 3 * C, device=dev).half()
attn_case("randn, 1 iter at a time", N, h, S, q_rand, 1)
attn_case("randn, 5 back-to-back", N, h, S, q_rand, 5)
attn_case("randn, 20 back-to-back", N, h, S, q_rand, 20)
attn_case("zeros, 20 back-to-back", N, h, S, torch.zeros_like(q_rand), 20)
attn_case("randn*0.3, 20 back-to-back", N, h, S, q_rand * 0.3, 20)
time.sleep(2.0)
attn_case("randn, 20 after 2 s idle", N, h, S, q_rand, 20)
attn_case("B=1 shape randn 20", 16, h, S, q_rand[: 16 * S], 20)
attn_case("S=1024 h=10 randn 20", 48, 10, 1024, torch.randn(48 * 1024, 3 * 640, device=dev).half(), 20)

T = 48 * 4096
H = 64
if "--gemm-only" in sys.argv:
    lines.clear()
for glds, flags in ((True, 0), (True, 8), (True, 4), (False, 0)):
    ops.GEMM_FLAGS = flags
    tag = f"[glds={int(glds)} flags={flags}] "
    gemm_case(tag + "conv3x3 320->320 @64x64", T, 320, 320, 10, mode=1, conv=(H, H, H, H, 1, 0, T), flags_glds=glds)
    gemm_case(tag + "linear 320->320", T, 320, 320, 20, flags_glds=glds)
    gemm_case(tag + "linear 320->960 (qkv)", T, 960, 320, 20, flags_glds=glds)
    gemm_case(tag + "linear 1280->320 (ff down)", T, 320, 1280, 10, flags_glds=glds)
    gemm_case(tag + "geglu 320->2560", T, 2560, 320, 10, flags_glds=glds, act=3)
    gemm_case(tag + "conv3x3 640->640 @32x32", T // 4, 640, 640, 10, mode=1, conv=(32, 32, 32, 32, 1, 0, T // 4), flags_glds=glds)
    gemm_case(tag + "conv3x3 1280->1280 @16x16", T // 16, 1280, 1280, 10, mode=1, conv=(16, 16, 16, 16, 1, 0, T // 16), flags_glds=glds)
    gemm_case(tag + "conv3x3 1280->1280 @8x8", T // 64, 1280, 1280, 10, mode=1, conv=(8, 8, 8, 8, 1, 0, T // 64), flags_glds=glds)
    gemm_case(tag + "linear 1280->1280 @16x16", T // 16, 1280, 1280, 20, flags_glds=glds)
ops.GEMM_FLAGS = 0

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "attn_probe.txt"), "w").write("\n".join(lines) + "\n")
