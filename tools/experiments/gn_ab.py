"""A/B of the one-launch GroupNorm (gn_fused_kernel) against the partial + apply pair (ANYV2V_GN_FUSED=0) on the GroupNorm launches
of the bench's two steps (B = 1 inversion, B = 3 edit).  Writes gpurun_out/gn_ab.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


# (tag, rows, C0, C1, rows per group, launches per step of that kind -- from profiles/r04_shape_report_B{1,3}.txt)
CASES = [
    ("B1 64x64 5-D", 65536, 320, 0, 65536, 26), ("B1 64x64 4-D", 65536, 320, 0, 4096, 13), ("B1 32x32 5-D", 16384, 640, 0, 16384, 25),
    ("B1 32x32 4-D", 16384, 640, 0, 1024, 10), ("B1 16x16 5-D", 4096, 1280, 0, 4096, 25), ("B1 16x16 4-D", 4096, 1280, 0, 256, 10),
    ("B1 8x8 5-D", 1024, 1280, 0, 1024, 28), ("B1 8x8 4-D", 1024, 1280, 0, 64, 11), ("B1 16x16 concat", 4096, 1280, 1280, 256, 3),
    ("B1 32x32 concat", 16384, 640, 640, 1024, 3), ("B1 64x64 concat", 65536, 320, 320, 4096, 3),
    ("B3 16x16 5-D", 12288, 1280, 0, 4096, 25), ("B3 16x16 4-D", 12288, 1280, 0, 256, 10), ("B3 8x8 5-D", 3072, 1280, 0, 1024, 28),
    ("B3 8x8 4-D", 3072, 1280, 0, 64, 11), ("B3 32x32 5-D (too big: same kernel pair both ways)", 49152, 640, 0, 16384, 25),
]
tot = [0.0, 0.0]
for tag, M, c0, c1, rpg, n in CASES:
    x0 = torch.randn(M, c0, device=dev).half()
    x1 = torch.randn(M, c1, device=dev).half() if c1 else None
    C = c0 + c1
    ga, be = torch.randn(C, device=dev).half(), torch.randn(C, device=dev).half()
    st = torch.zeros(ops.gn_scratch_floats(M, rpg), dtype=torch.float32, device=dev)
    out = torch.empty(M, C, dtype=torch.float16, device=dev)
    us = []
    for sw in ("0", "1"):
        os.environ["ANYV2V_GN_FUSED"] = sw
        us.append(timeit(lambda: ops.groupnorm(x0, ga, be, st, rpg, x1=x1, groups=32, silu=True, out=out)))
    os.environ.pop("ANYV2V_GN_FUSED")
    mb = 3 * M * C * 2 / 1e6
    lines.append(f"{tag:<52s} [{M:6d} x {C:4d}, {M // rpg:3d} groups]: two kernels {us[0]:6.1f} us ({mb / us[0]:5.2f} TB/s as 3 passes) | one launch {us[1]:6.1f} us"
                 f" | x{n} per step: {n * (us[0] - us[1]) / 1e3:+.3f} ms")
    print(lines[-1], flush=True)
    tot[0 if tag.startswith("B1") else 1] += n * (us[0] - us[1]) / 1e3
lines.append(f"sum over the listed launches: B = 1 step {tot[0]:+.3f} ms, B = 3 step {tot[1]:+.3f} ms (isolated launch times)")
print(lines[-1])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gn_ab.txt"), "w").write("\n".join(lines) + "\n")
