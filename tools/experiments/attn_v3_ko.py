import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
import torch
from anyv2v_amd import ops
def run(flags, qkv, o, N, h, S):
    ops.ATTN_FLAGS = flags
    C = 64 * h
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=N, heads=h, Sq=S, Sk=S, inner=1, q_strides=(S, 0, 1), kv_strides=(S, 0, 1))
    ops.ATTN_FLAGS = 0
def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
N, h, S = 48, 5, 4096
qkv = torch.randn(N * S, 3 * 64 * h, device="cuda").half()
o = torch.zeros(N * S, 64 * h, dtype=torch.float16, device="cuda")
V = [0] + [(k << 8) for k in (1, 2, 4, 8, 15)] + [32] + [32 | (k << 8) for k in (1, 2, 3, 4, 5, 8, 15)]
for _ in range(20): run(0, qkv, o, N, h, S)
best = {f: [] for f in V}
for r in range(4):
    for f in V:
        best[f].append(timeit(lambda: run(f, qkv, o, N, h, S), 10))
names = {0: "v2<3,1,8>", 32: "v3"}
out = []
for f in V:
    ko = f >> 8
    out.append(f"{names.get(f, ('v3' if f & 32 else 'v2<3,1,8>') + ' KO=%d' % ko):16s} min {min(best[f]):.4f} ms")
    print(out[-1], flush=True)
open(os.path.join(ROOT, "gpurun_out", "attn_v3_ko.txt"), "w").write("KO bits: 1 no range-check branches, 2 no barrier/vmcnt, 4 exp->mul, 8 no LDS fragment reads\n" + "\n".join(out) + "\n")
