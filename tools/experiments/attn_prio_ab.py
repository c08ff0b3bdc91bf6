"""A/B of the wave-priority variants of the graded attention kernel (descriptor flag bits 8-9, `flash_attn_d64_v2_prio_kernel`; apply tools/experiments/attn_prio_variants.patch and rebuild first):
0 = product kernel, 256 = s_setprio 1 around the MFMA clusters, 512 = static priority for waves 4-7, 768 = both.
Interleaved rounds at (48, 5, 4096, 64) and (16, 5, 4096, 64); outputs compared bit for bit.  Writes gpurun_out/attn_prio_ab.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []


def run(flags, q, o, N, h, S):
    ops.ATTN_FLAGS = flags
    C = 64 * h
    ops.attention(q[:, :C], q[:, C:2 * C], q[:, 2 * C:], o, batch=N, heads=h, Sq=S, Sk=S, inner=1, q_strides=(S, 0, 1), kv_strides=(S, 0, 1))


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


VARIANTS = (0, 256, 512, 768)
for N in (48, 16):
    h, S = 5, 4096
    q = torch.randn(N * S, 3 * 64 * h, device=dev).half()
    outs = {f: torch.empty(N * S, 64 * h, dtype=torch.float16, device=dev) for f in VARIANTS}
    for _ in range(30):          # clock ramp
        run(0, q, outs[0], N, h, S)
    torch.cuda.synchronize()
    for f in VARIANTS:
        run(f, q, outs[f], N, h, S)
    torch.cuda.synchronize()
    same = {f: bool(torch.equal(outs[f], outs[0])) for f in VARIANTS}
    best = {f: [] for f in VARIANTS}
    for r in range(5):
        for f in VARIANTS:
            best[f].append(timeit(lambda: run(f, q, outs[f], N, h, S), 20))
    for f in VARIANTS:
        ms = sorted(best[f])
        tf = 4.0 * N * h * S * S * 64 / (ms[0] * 1e-3) / 1e12
        lines.append(f"N={N:2d} flags={f:3d}: min {ms[0]:.4f} ms  median {ms[2]:.4f} ms  ({tf:6.1f} TFLOP/s at min, frac {tf / 2500:.4f})  bit-equal to flags 0: {same[f]}")
        print(lines[-1], flush=True)
ops.ATTN_FLAGS = 0
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "attn_prio_ab.txt"), "w").write("\n".join(lines) + "\n")
