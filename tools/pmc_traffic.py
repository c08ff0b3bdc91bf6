"""HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/rNN_traffic.json (read by bench.py).
    python tools/pmc_traffic.py <out.json> <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, hence x2 (MI355X_MICROARCH.md, HBM)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys


def means(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row.get("Kernel_Name", "?").split("(")[0]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


out, dfetch, dwrite = sys.argv[1:4]
fetch, write = means(dfetch, "FETCH_SIZE"), means(dwrite, "WRITE_SIZE")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:
    commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    commit = None
rec = {"commit": commit or os.environ.get("GIT_COMMIT"), "units": "FETCH_SIZE / WRITE_SIZE in KiB; hbm_bytes_per_launch = 2 * FETCH * 1024 + WRITE * 1024",
       "kernels": {}}
for k in sorted(set(fetch) & set(write)):
    f, nf = fetch[k]
    w, nw = write[k]
    rec["kernels"][k] = {"fetch_size_kib_mean": round(f, 1), "write_size_kib_mean": round(w, 1), "launches": [nf, nw],
                         "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024)}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
