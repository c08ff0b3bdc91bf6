"""GroupNorm / LayerNorm timing at the UNet's shapes (HBM-bound kernels). Writes gpurun_out/norm_probe.txt.
Traffic model: GroupNorm 2 reads + 1 write of X, LayerNorm 1 read + 1 write."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gn_case(M, C, rows, silu, two=False):
    C0 = C // 2 if two else C
    x0 = torch.randn(M, C0, device=dev).half()
    x1 = torch.randn(M, C - C0, device=dev).half() if two else None
    g = torch.randn(C, device=dev).half()
    b = torch.randn(C, device=dev).half()
    stats = torch.empty(ops.gn_scratch_floats(M, rows, 32), dtype=torch.float32, device=dev)
    out = torch.empty(M, C, dtype=torch.float16, device=dev)
    ms = timeit(lambda: ops.groupnorm(x0, g, b, stats, rows, x1=x1, silu=silu, out=out))
    gb = 3 * M * C * 2 / 1e9
    lines.append(f"groupnorm M={M:7d} C={C:5d} rows/group={rows:6d} silu={int(silu)} 2src={int(two)}: {ms * 1e3:8.1f} us  {gb / ms:7.2f} TB/s")
    print(lines[-1], flush=True)


def ln_case(M, C):
    x = torch.randn(M, C, device=dev).half()
    g = torch.randn(C, device=dev).half()
    b = torch.randn(C, device=dev).half()
    out = torch.empty_like(x)
    ms = timeit(lambda: ops.layernorm(x, g, b, out=out))
    gb = 2 * M * C * 2 / 1e9
    lines.append(f"layernorm M={M:7d} C={C:5d}: {ms * 1e3:8.1f} us  {gb / ms:7.2f} TB/s")
    print(lines[-1], flush=True)


for _ in range(2):  # first pass doubles as the clock warm-up
    lines.clear()
    gn_case(196608, 320, 65536, True)
    gn_case(196608, 320, 4096, True)
    gn_case(196608, 320, 4096, False)
    gn_case(49152, 640, 16384, True)
    gn_case(49152, 640, 1024, True)
    gn_case(12288, 1280, 4096, True)
    gn_case(3072, 1280, 1024, True)
    gn_case(196608, 640, 4096, True, two=True)
    gn_case(196608, 960, 4096, True, two=True)
    gn_case(65536, 320, 65536, True)
    gn_case(65536, 320, 4096, True)
    ln_case(196608, 320)
    ln_case(49152, 640)
    ln_case(12288, 1280)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "norm_probe.txt"), "w").write("\n".join(lines) + "\n")
