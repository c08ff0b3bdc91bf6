"""Parity of the ping-pong GEMM kernel (gemm_pp_kernel, flags bit10) on the GPU: python tools/gemm_pp_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_checks as gc  # noqa: E402

res = gc.check_gemm_pp()
for r in res:
    print(f"{'ok  ' if r['ok'] else 'FAIL'} {r['name']:<60s} err {r['err']:.3e} (tol {r['tol']:.1e})", flush=True)
sys.exit(0 if all(r["ok"] for r in res) else 1)
