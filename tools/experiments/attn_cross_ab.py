"""A/B of descriptor-flag variants on the cross-attention launches (Sk = 145, K / V shared by the 16 frames of a clip) and the short
spatial launches.  VARIANTS env.  Writes gpurun_out/attn_cross_ab.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,512").split(",")]
lines = []


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for (B, Fr, h, S, Sk) in [tuple(int(x) for x in c.split(":")) for c in os.environ.get("CASES", "3:16:5:4096:145,1:16:5:4096:145,3:16:10:1024:145,3:16:20:256:145,3:16:20:256:256").split(",")]:
    N, C = B * Fr, 64 * h
    q = torch.randn(N * S, C, device="cuda").half()
    self_attn = Sk == S
    kv = torch.randn((N if self_attn else B) * Sk, 2 * C, device="cuda").half()
    outs = {f: torch.zeros(N * S, C, dtype=torch.float16, device="cuda") for f in VARIANTS}

    def run(f):
        ops.ATTN_FLAGS = f
        ops.attention(q, kv[:, :C], kv[:, C:], outs[f], batch=N, heads=h, Sq=S, Sk=Sk, inner=1, q_strides=(S, 0, 1), kv_strides=(Sk, 0, 1),
                      kv_div=1 if self_attn else Fr)
        ops.ATTN_FLAGS = 0
    for f in VARIANTS:
        for _ in range(5):
            run(f)
    torch.cuda.synchronize()
    best = {f: [] for f in VARIANTS}
    for r in range(5):
        for f in VARIANTS:
            best[f].append(timeit(lambda: run(f), 20))
    for f in VARIANTS:
        ms = sorted(best[f])
        lines.append(f"N={N:2d} h={h:2d} Sq={S} Sk={Sk} flags={f:3d}: min {ms[0] * 1e3:7.1f} us median {ms[2] * 1e3:7.1f} us  bit-equal: {bool(torch.equal(outs[f], outs[VARIANTS[0]]))}")
        print(lines[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "attn_cross_ab.txt"), "w").write("\n".join(lines) + "\n")
