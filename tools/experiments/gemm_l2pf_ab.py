"""A/B of the cross-tile L2 prefetch of the persistent GEMM kernel (descriptor flag bit21) on the short-K linear / GEGLU launches of the
32x32 and 16x16 levels (B = 3 and B = 1).  Writes gpurun_out/gemm_l2pf_ab.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []
PF = 1 << 21


def timeit(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


CASES = [("L1 QKV", 49152, 1920, 640, 0, False), ("L1 out-proj +res", 49152, 640, 640, 0, True), ("L1 to_q", 49152, 640, 640, 0, False),
         ("L1 GEGLU", 49152, 5120, 640, 3, False), ("L1 FF down +res", 49152, 640, 2560, 0, True),
         ("L2 QKV", 12288, 3840, 1280, 0, False), ("L2 out-proj +res", 12288, 1280, 1280, 0, True), ("L2 GEGLU", 12288, 10240, 1280, 3, False),
         ("L2 FF down +res", 12288, 1280, 5120, 0, True), ("B1 L1 QKV", 16384, 1920, 640, 0, False), ("B1 L1 GEGLU", 16384, 5120, 640, 3, False),
         ("B1 L1 out-proj +res", 16384, 640, 640, 0, True)]
tot = [0.0, 0.0]
for tag, M, N, K, act, res in CASES:
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.zeros(N, dtype=torch.float16, device=dev)
    n_out = N // 2 if act == 3 else N
    r = torch.randn(M, n_out, device=dev).half() if res else None
    outs = {}
    us = {0: [], PF: []}
    for f in (0, PF):
        outs[f] = torch.empty(M, n_out, dtype=torch.float16, device=dev)
        ops.GEMM_FLAGS = f
        for _ in range(3):
            ops.gemm(a, w, bias=b, out=outs[f], act=act, residual=r, M=M)
    torch.cuda.synchronize()
    for rep in range(5):
        for f in (0, PF):
            ops.GEMM_FLAGS = f
            us[f].append(timeit(lambda: ops.gemm(a, w, bias=b, out=outs[f], act=act, residual=r, M=M)))
    ops.GEMM_FLAGS = 0
    m0, m1 = min(us[0]), min(us[PF])
    fl = 2.0 * M * N * K
    lines.append(f"{tag:<20s} M={M:6d} N={N:5d} K={K:5d}: plain {m0:7.1f} us ({fl / m0 / 1e6:5.0f} TF) | prefetch {m1:7.1f} us ({fl / m1 / 1e6:5.0f} TF) "
                 f"| {100 * (m1 / m0 - 1):+.1f} %  bit-equal {bool(torch.equal(outs[0], outs[PF]))}")
    print(lines[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_l2pf_ab.txt"), "w").write("\n".join(lines) + "\n")
