"""The K = 320 launches of the 64x64 level, a few times each, for `rocprofv3 --pmc ...` passes over gemm_ws_kernel (and, with
WS_FLAGS=512, the tile kernels it replaces).  python tools/ws_pmc_targets.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ops.GEMM_FLAGS = int(os.environ.get("WS_FLAGS", "0"))
K, M = 320, 196608
a = torch.randn(M, K, device="cuda").half()
r = torch.randn(M, 320, device="cuda").half()
for (N, act, res) in [(320, 0, False), (320, 0, True), (960, 0, False), (2560, 3, False)]:
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device="cuda")
    out = torch.empty(M, N // 2 if act == 3 else N, dtype=torch.float16, device="cuda")
    for _ in range(reps):
        ops.gemm(a, w, bias=b, out=out, act=act, residual=r if res else None)
torch.cuda.synchronize()
print("done")
