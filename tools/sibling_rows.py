"""Prints the pipeline-level rows of the two sibling backends (ConsistI2V, SEINE) with their calibration figures:
HIP vs the reference class's fp32 output, the torch op emulation (independent fp16-storage run) vs the same, HIP vs emulation.
Writes gpurun_out/sibling_rows.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_checks as gc  # noqa: E402

rows = gc.check_consisti2v_pipeline() + gc.check_seine_pipeline()
lines = [f"{'ok  ' if r['ok'] else 'FAIL'} {r['name']}: {r['err']:.3e} (tol {r['tol']:.2e})" for r in rows]
print("\n".join(lines))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "sibling_rows.txt"), "w").write("\n".join(lines) + "\n")
