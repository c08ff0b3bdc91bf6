"""A/B of the temporal (16-frame) attention launch under PnP injection: shared-softmax form (one wave per source sequence, flags 0) vs
per-branch aliasing (flags 8).  Writes gpurun_out/attn_temporal_ab.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

lines = []


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for (B, Fr, HW, h) in ((3, 16, 4096, 5), (3, 16, 1024, 10), (3, 16, 256, 20), (3, 16, 64, 20)):
    C = 64 * h
    qkv = torch.randn(B * Fr * HW, 3 * C, device="cuda").half()
    st = (Fr * HW, 1, HW)
    outs = {f: torch.zeros(B * Fr * HW, C, dtype=torch.float16, device="cuda") for f in (8, 0)}

    def run(f):
        ops.ATTN_FLAGS = f
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], outs[f], batch=B * HW, heads=h, Sq=Fr, Sk=Fr, inner=HW, q_strides=st,
                      kv_strides=st, qk_mod=HW)
        ops.ATTN_FLAGS = 0
    for f in (8, 0):
        for _ in range(5):
            run(f)
    torch.cuda.synchronize()
    best = {8: [], 0: []}
    for r in range(5):
        for f in (8, 0):
            best[f].append(timeit(lambda: run(f), 20))
    a, s = min(best[8]) * 1e3, min(best[0]) * 1e3
    lines.append(f"temporal PnP B={B} F={Fr} HW={HW} heads={h}: aliasing {a:7.1f} us | shared softmax {s:7.1f} us | {100 * (s / a - 1):+.1f} %  bit-equal "
                 f"{bool(torch.equal(outs[0], outs[8]))}")
    print(lines[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "attn_temporal_ab.txt"), "w").write("\n".join(lines) + "\n")
