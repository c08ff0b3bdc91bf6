"""Parity + A/B of the software-pipelined spatial-attention kernel (flash_attn_d64_v3_kernel; descriptor flag bit5 = 8-wave blocks,
bit6 = 4-wave blocks) against the v2 kernels the dispatcher takes by default (flags 0) and against fp32 SDPA.

1. every plain-attention case of tests/gpu_checks.py::check_attention / check_attention_forced_rescale (ragged, short, spiked inputs
   that force the rescale path at every step position) with the v3 kernels forced;
2. bit-equality of v3 and v2 outputs on bounded random data (same products, same order: the pipeline only moves issue slots);
3. interleaved timing rounds at the graded shape (48, 5, 4096, 64), the inversion step's (16, 5, 4096, 64) and the 32x32 level.
Writes gpurun_out/attn_v3_ab.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []


def say(s):
    lines.append(s)
    print(s, flush=True)


def run(flags, qkv, o, N, h, S, **kw):
    saved, ops.ATTN_FLAGS = ops.ATTN_FLAGS, flags
    try:
        C = 64 * h
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=N, heads=h, Sq=S, Sk=S, inner=1, q_strides=(S, 0, 1),
                      kv_strides=(S, 0, 1), **kw)
    finally:
        ops.ATTN_FLAGS = saved


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,32,64").split(",")]

# ---- 1. the checks of the GPU suite with the v3 kernels forced
import gpu_checks  # noqa: E402

for f in [v for v in VARIANTS if v]:
    ops.ATTN_FLAGS = f
    try:
        res = gpu_checks.check_attention_forced_rescale() + gpu_checks.check_attention(naive_too=False)
    finally:
        ops.ATTN_FLAGS = 0
    bad = [r for r in res if not r["ok"]]
    worst = max(res, key=lambda r: (r["err"] if r["err"] == r["err"] else 1e9) / max(r["tol"], 1e-30))
    say(f"[flags {f}] gpu_checks attention rows: {len(res) - len(bad)}/{len(res)} ok; worst {worst['name']}: {worst['err']:.3e} (tol {worst['tol']:.1e})")
    for r in bad:
        say(f"    FAIL {r['name']}: {r['err']:.3e} (l2 {r['l2']:.3e}) > {r['tol']:.1e}")
    for r in res:
        if "forced rescale" in r["name"]:
            say(f"    {r['name']}: {r['err']:.3e}")

# ---- 2. + 3. bit-equality and timing
for (N, h, S, tag) in ((48, 5, 4096, "graded: spatial 64x64, 3 branches"), (16, 5, 4096, "spatial 64x64, 1 branch (inversion step)"),
                       (48, 10, 1024, "spatial 32x32, 3 branches"), (16, 10, 1024, "spatial 32x32, 1 branch")):
    qkv = torch.randn(N * S, 3 * 64 * h, device=dev).half()
    outs = {f: torch.zeros(N * S, 64 * h, dtype=torch.float16, device=dev) for f in VARIANTS}
    for _ in range(20):          # clock ramp
        run(0, qkv, outs[0], N, h, S)
    torch.cuda.synchronize()
    for f in VARIANTS:
        run(f, qkv, outs[f], N, h, S)
    torch.cuda.synchronize()
    same = {f: bool(torch.equal(outs[f], outs[VARIANTS[0]])) for f in VARIANTS}
    md = {f: float((outs[f].float() - outs[VARIANTS[0]].float()).abs().max()) for f in VARIANTS}
    best = {f: [] for f in VARIANTS}
    for r in range(5):
        for f in VARIANTS:
            best[f].append(timeit(lambda: run(f, qkv, outs[f], N, h, S), 20))
    for f in VARIANTS:
        ms = sorted(best[f])
        tf = 4.0 * N * h * S * S * 64 / (ms[0] * 1e-3) / 1e12
        say(f"{tag}: N={N:2d} h={h} S={S} flags={f:3d}: min {ms[0]:.4f} ms  median {ms[2]:.4f} ms  ({tf:6.1f} TFLOP/s at min, frac {tf / 2500:.4f})"
            f"  bit-equal to flags {VARIANTS[0]}: {same[f]} (max abs diff {md[f]:.2e})")
    del qkv, outs
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "attn_v3_ab.txt"), "w").write("\n".join(lines) + "\n")
