"""Launch the bench's roofline kernels a few times each so that a `rocprofv3 --pmc ...` pass stays short: spatial self-attention at
the graded shape (N=48, h=5, S=4096, d=64; plain and PnP shared-softmax launch) and the conv3x3 320->320 @64x64 (N=48).
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o t -- python tools/pmc_targets.py
(one pass per counter group; summarise with tools/pmc_traffic.py / tools/pmc_summary.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, h, S = 48, 5, 4096
C = 64 * h
q = torch.randn(N * S, 3 * C, device="cuda").half()
o = torch.empty(N * S, C, dtype=torch.float16, device="cuda")
for qk_mod in (0, N // 3):
    for _ in range(reps):
        ops.attention(q[:, :C], q[:, C:2 * C], q[:, 2 * C:], o, batch=N, heads=h, Sq=S, Sk=S, inner=1,
                      q_strides=(S, 0, 1), kv_strides=(S, 0, 1), qk_mod=qk_mod)
H = 64
x = torch.randn(N * H * H, C, device="cuda").half()
w = (torch.randn(C, 9 * C, device="cuda") / (9 * C) ** 0.5).half()
b = torch.zeros(C, dtype=torch.float16, device="cuda")
out = torch.empty(N * H * H, C, dtype=torch.float16, device="cuda")
for flags in [int(f) for f in os.environ.get("PMC_GEMM_FLAGS", "0").split(",")]:  # e.g. "0,512": both K orders of the conv gather
    ops.GEMM_FLAGS = flags
    for _ in range(reps):
        ops.gemm(x, w, bias=b, mode=ops.MODE_CONV2D, conv=(H, H, H, H, 1, 0), out=out)
ops.GEMM_FLAGS = 0
# weight-stationary GEMMs of the 64x64 level at the edit step's row count: feed-forward up-projection + GEGLU and fused QKV
M, K = 196608, C
xa = torch.randn(M, K, device="cuda").half()
for Nn, act in ((2560, ops.ACT_GEGLU), (960, ops.ACT_NONE)):
    ww = (torch.randn(Nn, K, device="cuda") / K ** 0.5).half()
    bb = torch.zeros(Nn, dtype=torch.float16, device="cuda")
    oo = torch.empty(M, Nn // 2 if act == ops.ACT_GEGLU else Nn, dtype=torch.float16, device="cuda")
    for _ in range(reps):
        ops.gemm(xa, ww, bias=bb, act=act, out=oo)
# round 4: the rastered GEGLU launches of the persistent kernel (640 / 1280 channels) and the fused feed-forward kernel (320 channels)
for (Mg, Ng, Kg) in ((49152, 5120, 640), (12288, 10240, 1280)):
    xg = torch.randn(Mg, Kg, device="cuda").half()
    wg = (torch.randn(Ng, Kg, device="cuda") / Kg ** 0.5).half()
    bg = torch.zeros(Ng, dtype=torch.float16, device="cuda")
    og = torch.empty(Mg, Ng // 2, dtype=torch.float16, device="cuda")
    for _ in range(reps):
        ops.gemm(xg, wg, bias=bg, act=ops.ACT_GEGLU, out=og)
w1 = (torch.randn(2560, 320, device="cuda") / 320 ** 0.5).half()
b1 = torch.zeros(2560, dtype=torch.float16, device="cuda")
w2s = ops.ff_pack_w2((torch.randn(320, 1280, device="cuda") / 1280 ** 0.5).half())
b2 = torch.zeros(320, dtype=torch.float16, device="cuda")
yo = torch.empty(M, 320, dtype=torch.float16, device="cuda")
for _ in range(reps):
    ops.ff_geglu(xa, w1, b1, w2s, b2, residual=xa, out=yo)
torch.cuda.synchronize()
print("done")
