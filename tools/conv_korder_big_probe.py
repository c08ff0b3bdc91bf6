"""Probe (experiments build): conv K order in gemm_big_kernel.  flags bits 29-30: 0 = tap-major (product), 1 = tap-major with the gather
addresses recomputed every K-tile (the cost of the recomputation alone), 2 = slice-major (slice, tap) with the same recomputation.  If 2 is
faster than 1, the order itself pays (L2 reuse of the A rows across the nine taps) and a cheap incremental address update would keep it.
gpurun_out/r06_conv_korder_big.txt
(the probe orders live in tools/experiments/conv_korder_big_probe.patch, against the tree of commit b0a3c05..: apply, `make experiments`, run)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "libanyv2v_hip_experiments.so")
from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []
ARMS = (("tap-major", 8), ("tap-major + recompute", 8 | (1 << 29)), ("slice-major + recompute", 8 | (2 << 29)))
for (tag, n_img, H, cin, cout) in [("B3 64x64 320->320", 48, 64, 320, 320), ("B3 64x64 640->320", 48, 64, 640, 320), ("B3 64x64 960->320", 48, 64, 960, 320),
                                   ("B3 32x32 640->640", 48, 32, 640, 640), ("B3 32x32 1280->640", 48, 32, 1280, 640), ("B3 16x16 1280->1280", 48, 16, 1280, 1280),
                                   ("B3 16x16 2560->1280", 48, 16, 2560, 1280), ("B1 64x64 320->320", 16, 64, 320, 320)]:
    M, K = n_img * H * H, 9 * cin
    x = torch.randn(M, cin, device=dev).half()
    w = (torch.randn(cout, K, device=dev) / K ** 0.5).half()
    b = torch.randn(cout, device=dev).half()
    outs = [torch.empty(M, cout, dtype=torch.float16, device=dev) for _ in ARMS]
    times = [[] for _ in ARMS]
    kw = dict(bias=b, mode=ops.MODE_CONV2D, conv=(H, H, H, H, 1, 0))
    for i, (_, fl) in enumerate(ARMS):
        ops.GEMM_FLAGS = fl
        for _ in range(2):
            ops.gemm(x, w, out=outs[i], **kw)
    torch.cuda.synchronize()
    for _ in range(6):
        for i, (_, fl) in enumerate(ARMS):
            ops.GEMM_FLAGS = fl
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm(x, w, out=outs[i], **kw)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 4 * 1e3)
    ops.GEMM_FLAGS = 0
    fl_ = 2.0 * M * cout * K
    ref = outs[0].float()
    errs = [float((o.float() - ref).abs().max() / ref.abs().max()) for o in outs]
    row = f"{tag:<22s}: " + " | ".join(f"{n} {sorted(t)[len(t) // 2]:7.1f} us ({fl_ / sorted(t)[len(t) // 2] / 1e6:5.0f} TF, err {e:.1e})" for (n, _), t, e in zip(ARMS, times, errs))
    lines.append(row)
    print(row, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r06_conv_korder_big.txt"), "w").write("\n".join(lines) + "\n")
