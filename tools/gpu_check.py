"""Run every GPU parity check and write a report (gpurun_out/gpu_check.json + .txt). Diagnostic twin of `pytest -m gpu`.

    python tools/gpu_check.py [kernels] [mini] [loops] [full]
"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import gpu_checks as gc  # noqa: E402


def main():
    what = set(sys.argv[1:]) or {"kernels", "mini", "loops"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    report = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "results": []}
    groups = []
    for f in gc.ALL_KERNEL_CHECKS:
        if "kernels" in what or f.__name__.replace("check_", "") in what:
            groups.append((f.__name__, f))
    if "mini" in what:
        groups += [("unet_golden", gc.check_unet_golden),
                   ("unet_mini_vs_oracle", lambda: gc.check_unet_vs_oracle("mini", 3, 4, 8, report=report)),
                   ("unet_mini_B1", lambda: gc.check_unet_vs_oracle("mini", 1, 8, 16, with_pnp=False, report=report))]
    if "loops" in what:
        groups += [("loops_mini", gc.check_loops_mini)]
    if "full" in what:
        groups += [("unet_full_config1_B1", lambda: gc.check_unet_vs_oracle("full", 1, 8, 32, with_pnp=False, report=report)),
                   ("unet_full_config1_B3", lambda: gc.check_unet_vs_oracle("full", 3, 8, 32, with_pnp=True, report=report))]
    lines = []
    nfail = 0
    for name, fn in groups:
        t0 = time.time()
        try:
            res = fn()
            torch.cuda.synchronize()
        except Exception:
            tb = traceback.format_exc()
            res = [dict(name=name + " EXCEPTION", err=float("nan"), tol=0, ok=False, trace=tb)]
            lines.append(tb)
        for r in res:
            ok = r["ok"] or r.get("informational", False)
            nfail += 0 if ok else 1
            lines.append(f"{'PASS' if r['ok'] else ('INFO' if r.get('informational') else 'FAIL')}  {r['name']:<70s} "
                         f"max-rel {r['err']:.3e}  l2 {r.get('l2', float('nan')):.3e}  tol {r['tol']:.1e}")
            if "dump" in r:
                lines.append("      dump: " + json.dumps(r["dump"]))
        report["results"] += [{k: v for k, v in r.items()} for r in res]
        lines.append(f"---- {name}: {time.time() - t0:.1f}s")
        print("\n".join(lines[-(len(res) + 1):]), flush=True)
    report["failures"] = nfail
    tag = "_".join(sorted(what))
    with open(os.path.join(ROOT, "gpurun_out", f"gpu_check_{tag}.json"), "w") as f:
        json.dump(report, f, indent=1, default=str)
    with open(os.path.join(ROOT, "gpurun_out", f"gpu_check_{tag}.txt"), "w") as f:
        f.write("\n".join(lines) + f"\nFAILURES: {nfail}\n")
    print(f"FAILURES: {nfail}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
