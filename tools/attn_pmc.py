"""Launch the spatial self-attention kernels a few times at the graded shape (N=48, h=5, S=4096, d=64) so that a
`rocprofv3 --pmc ...` pass stays short:  plain launch and PnP shared-softmax launch (+ any variant flags named in ATTN_PMC_FLAGS).
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -d gpurun_out/pmc -o attn -- python tools/attn_pmc.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

N, h, S = 48, 5, 4096
C = 64 * h
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
q = torch.randn(N * S, 3 * C, device="cuda").half()
o = torch.empty(N * S, C, dtype=torch.float16, device="cuda")
extra = tuple((int(f), 0) for f in os.environ.get("ATTN_PMC_FLAGS", "").split(",") if f)
for flags, qk_mod in ((0, 0),) + extra + ((0, N // 3),):  # plain, variants under test, PnP shared softmax
    ops.ATTN_FLAGS = flags
    for _ in range(reps):
        ops.attention(q[:, :C], q[:, C:2 * C], q[:, 2 * C:], o, batch=N, heads=h, Sq=S, Sk=S, inner=1,
                      q_strides=(S, 0, 1), kv_strides=(S, 0, 1), qk_mod=qk_mod)
torch.cuda.synchronize()
print("done")
