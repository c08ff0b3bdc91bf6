"""Probe (experiments build): does a slice-major conv K order (channel slice, dy, dx) turn the dx re-reads of the A operand into L1
hits?  gemm_sw_kernel conv2d launches in three forms: tap-major order (product), slice-major order with plain W pieces (KO 8), slice-major
with the W pieces past L1 (`sc1`, KO 7); gemm_big_kernel beside them.  Results of the two orders differ in the last bits only (fp32
accumulation order); checked against the tap-major result with a tolerance.  gpurun_out/r06_conv_korder_l1.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import _lib  # noqa: E402

if "--product" not in sys.argv:
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "libanyv2v_hip_experiments.so")
from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []
ARMS = (("gemm_big", 8), ("sw tap-major", 1 << 21), ("sw slice-major", (1 << 21) | (8 << 23)), ("sw slice-major + W sc1", (1 << 21) | (7 << 23)))
if "--product" in sys.argv:   # the LDS-patch kernel (gemm_swh.hip, flags bit28) against the default dispatch and the plain one-wave kernel
    ARMS = (("default", 0), ("sw tap-major", 1 << 21), ("swh LDS patch (dy, slice, dx)", 1 << 28))
for (tag, n_img, H, cin, cout) in [("B3 64x64 320->320", 48, 64, 320, 320), ("B3 64x64 640->320", 48, 64, 640, 320), ("B3 32x32 640->640", 48, 32, 640, 640),
                                   ("B3 16x16 1280->1280", 48, 16, 1280, 1280), ("B1 64x64 320->320", 16, 64, 320, 320), ("B3 32x32 1280->640", 48, 32, 1280, 640),
                                   ("B3 64x64 960->320", 48, 64, 960, 320), ("B3 16x16 2560->1280", 48, 16, 2560, 1280)]:
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
open(os.path.join(ROOT, "gpurun_out", "r06_conv_halo_ab.txt" if "--product" in sys.argv else "r06_conv_korder_l1.txt"), "w").write("\n".join(lines) + "\n")
