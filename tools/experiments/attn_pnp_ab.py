"""A/B of descriptor-flag variants of the shared-softmax (PnP injection) attention launch: flash_attn_d64_v2_kernel<2,3,4[,VPF]>.
VARIANTS env: comma-separated flag values (0 = product).  Parity rows of gpu_checks.check_attention with each flag, then interleaved
timing at (48, 5, 4096) and (48, 10, 1024) with qk_mod = 16.  Writes gpurun_out/attn_pnp_ab.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402
import gpu_checks  # noqa: E402

VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,256").split(",")]
lines = []


def say(s):
    lines.append(s)
    print(s, flush=True)


for f in [v for v in VARIANTS if v]:
    ops.ATTN_FLAGS = f
    try:
        res = gpu_checks.check_attention_forced_rescale() + gpu_checks.check_attention(naive_too=False)
    finally:
        ops.ATTN_FLAGS = 0
    bad = [r for r in res if not r["ok"]]
    say(f"[flags {f}] attention rows: {len(res) - len(bad)}/{len(res)} ok" + "".join(f"\n    FAIL {r['name']}: {r['err']:.3e}" for r in bad))


def run(flags, qkv, o, N, h, S):
    ops.ATTN_FLAGS = flags
    C = 64 * h
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=N, heads=h, Sq=S, Sk=S, inner=1, q_strides=(S, 0, 1),
                  kv_strides=(S, 0, 1), qk_mod=N // 3)
    ops.ATTN_FLAGS = 0


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for (N, h, S) in ((48, 5, 4096), (48, 10, 1024)):
    qkv = torch.randn(N * S, 3 * 64 * h, device="cuda").half()
    outs = {f: torch.zeros(N * S, 64 * h, dtype=torch.float16, device="cuda") for f in VARIANTS}
    for _ in range(20):
        run(0, qkv, outs[VARIANTS[0]], N, h, S)
    for f in VARIANTS:
        run(f, qkv, outs[f], N, h, S)
    torch.cuda.synchronize()
    best = {f: [] for f in VARIANTS}
    for r in range(5):
        for f in VARIANTS:
            best[f].append(timeit(lambda: run(f, qkv, outs[f], N, h, S), 20))
    for f in VARIANTS:
        ms = sorted(best[f])
        say(f"PnP shared softmax N={N} h={h} S={S} flags={f:3d}: min {ms[0]:.4f} ms median {ms[2]:.4f} ms  bit-equal to flags {VARIANTS[0]}: "
            f"{bool(torch.equal(outs[f], outs[VARIANTS[0]]))}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "attn_pnp_ab.txt"), "w").write("\n".join(lines) + "\n")
