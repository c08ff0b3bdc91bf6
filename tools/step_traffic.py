"""Whole-step HBM traffic from the two rocprofv3 --pmc passes over tools/step_traffic_target.py -> profiles/rNN_step_traffic.json.
    python tools/step_traffic.py <out.json> <FETCH_SIZE dir> <WRITE_SIZE dir> <pairs>
Every kernel dispatch of the passes is summed (FETCH_SIZE x 2: gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md HBM) and
divided by the number of step pairs; the engines' once-per-clip conditioning launches (< 0.5 % of the bytes) are included."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys


def total(d, counter):
    s, n, per = 0.0, 0, collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                v = float(row["Counter_Value"])
                s += v
                n += 1
                per[row.get("Kernel_Name", "?").split("(")[0][-60:]] += v
    return s, n, per


out, dfetch, dwrite, pairs = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
f, nf, pf = total(dfetch, "FETCH_SIZE")
w, nw, pw = total(dwrite, "WRITE_SIZE")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("GIT_COMMIT")
gb = lambda kib: kib * 1024 / 1e9
rec = {"commit": commit, "pairs": pairs, "dispatches": [nf, nw],
       "fetch_gb_per_step_pair": round(2 * gb(f) / pairs, 2), "write_gb_per_step_pair": round(gb(w) / pairs, 2),
       "hbm_gb_per_step_pair": round((2 * gb(f) + gb(w)) / pairs, 2),
       "unfused_estimate_gb_per_pnp_step": 152.0,
       "top_kernels_gb_per_pair": {k: round((2 * gb(pf.get(k, 0.0)) + gb(pw.get(k, 0.0))) / pairs, 2)
                                    for k in sorted(set(pf) | set(pw), key=lambda k: -(2 * pf.get(k, 0.0) + pw.get(k, 0.0)))[:12]},
       "units": "FETCH_SIZE / WRITE_SIZE in KiB summed over every dispatch of `python tools/step_traffic_target.py <pairs>` (eager launches, "
                "separate --pmc passes); hbm = 2 x FETCH + WRITE; a step pair = 1 inversion step (B=1) + 1 PnP edit step (B=3) at 16 f x 512^2"}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
