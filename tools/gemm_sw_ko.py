"""Knock-out series of the one-wave-per-SIMD kernel's K loop (probe build: make -C anyv2v_amd/csrc experiments): what a K-tile costs
without its LDS-DMA pieces (flags bits 23-24 = 1), without pieces and fragment reads (2), with pieces and reads but no MFMAs (3).
Results are garbage by construction; only the times mean something.  gpurun_out/r06_gemm_sw_knockouts.txt"""
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
ARMS = (("gemm_big (default)", 8), ("sw", 1 << 21), ("sw no DMA", (1 << 21) | (1 << 23)), ("sw no DMA no reads", (1 << 21) | (2 << 23)),
        ("sw no MFMA", (1 << 21) | (3 << 23)), ("sw W pieces only", (1 << 21) | (4 << 23)), ("sw A from zero line", (1 << 21) | (5 << 23)),
        ("sw A pieces every 3rd K-tile", (1 << 21) | (6 << 23)))
for (M, N, K) in [(12288, 1280, 11520), (12288, 1280, 5120), (49152, 640, 5760), (196608, 320, 2880), (49152, 1920, 640)]:
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.randn(N, device=dev).half()
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    times = [[] for _ in ARMS]
    for i, (_, fl) in enumerate(ARMS):
        ops.GEMM_FLAGS = fl
        for _ in range(2):
            ops.gemm(a, w, bias=b, out=out)
    torch.cuda.synchronize()
    for _ in range(6):
        for i, (_, fl) in enumerate(ARMS):
            ops.GEMM_FLAGS = fl
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm(a, w, bias=b, out=out)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 4 * 1e3)
    ops.GEMM_FLAGS = 0
    flops = 2.0 * M * N * K
    row = f"M={M:6d} N={N:5d} K={K:5d}: " + " | ".join(f"{n} {sorted(t)[len(t) // 2]:7.1f} us ({flops / sorted(t)[len(t) // 2] / 1e6:5.0f} TF)" for (n, _), t in zip(ARMS, times))
    lines.append(row)
    print(row, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r06_gemm_sw_knockouts.txt"), "w").write("\n".join(lines) + "\n")
