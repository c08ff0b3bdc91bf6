"""A few launches of the persistent tile GEMM on long-K shapes for an L2 counter pass:
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d <dir> -o t -- python tools/l2_hit_targets.py
(are the A / W tiles that several blocks of an XCD share served by its L2?)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

for (M, N, K, mode, conv) in [(12288, 1280, 11520, 0, None), (49152, 640, 5760, 0, None), (196608, 320, 2880, 1, (64, 64, 64, 64, 1, 0)),
                              (49152, 640, 5760, 1, (32, 32, 32, 32, 1, 0)), (12288, 1280, 11520, 1, (16, 16, 16, 16, 1, 0)), (49152, 5120, 640, 0, None)]:
    cin = K // 9 if mode == 1 else K
    a = torch.randn(M, cin, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device="cuda")
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    for flags in (0, 1 << 21):
        ops.GEMM_FLAGS = flags
        for _ in range(2):
            ops.gemm(a, w, bias=b, mode=mode, conv=conv, out=out)
ops.GEMM_FLAGS = 0
torch.cuda.synchronize()
print("done")
