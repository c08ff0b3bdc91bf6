"""Probe (experiments build): is the one-wave kernel's K loop bound by a per-CU resource or by one the CUs share (L2 / fabric)?
The same launches with 256, 128, 64 and 32 persistent blocks (ANYV2V_SW_GRID): the tile count is fixed, so time x blocks is the
CU-time the job costs.  Per-CU bound: CU-time constant.  Shared-resource bound: CU-time falls as blocks get fewer.
gpurun_out/r06_gemm_sw_grid_probe.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from anyv2v_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "libanyv2v_hip_experiments.so")
    from anyv2v_amd import ops
    ops.GEMM_FLAGS = 1 << 21
    for (M, N, K) in [(12288, 1280, 11520), (49152, 640, 5760), (196608, 320, 2880), (49152, 5120, 640)]:
        a = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        b = torch.randn(N, device="cuda").half()
        out = torch.empty(M, N, dtype=torch.float16, device="cuda")
        for _ in range(2):
            ops.gemm(a, w, bias=b, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.gemm(a, w, bias=b, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3 * 1e3)
        print(f"RESULT {M} {N} {K} {sorted(ts)[2]:.1f}", flush=True)
    sys.exit(0)
rows = {}
for g in (256, 128, 64, 32):
    env = dict(os.environ, ANYV2V_SW_GRID=str(g))
    out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout
    for line in out.splitlines():
        if line.startswith("RESULT"):
            _, M, N, K, us = line.split()
            rows.setdefault((int(M), int(N), int(K)), {})[g] = float(us)
lines = []
for (M, N, K), r in rows.items():
    fl = 2.0 * M * N * K
    lines.append(f"M={M:6d} N={N:5d} K={K:5d}: " + " | ".join(f"{g:3d} blocks {r[g]:8.1f} us = {r[g] * g / 256:8.1f} us x 256-CU-equivalents ({fl / r[g] / 1e6 * 256 / g:5.0f} TF per 256 CUs)" for g in sorted(r, reverse=True)))
    print(lines[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r06_gemm_sw_grid_probe.txt"), "w").write("\n".join(lines) + "\n")
