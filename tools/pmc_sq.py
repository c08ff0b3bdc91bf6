"""MFMA utilisation per kernel from a rocprofv3 SQ counter pass -> profiles/rNN_pmc.json (read by bench.py: `mfma_busy_frac`).
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \\
              SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d <dir> -o t -- python tools/pmc_targets.py
    python tools/pmc_sq.py <out.json> <dir>
Units (MI355X_MICROARCH.md, constants table): SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the 1024 SIMDs (16 per
16x16x32 f16 MFMA, 32 per 32x32x16); GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_BUSY_CYCLES over the 32 shader engines (32 SIMDs each).
    mfma_busy_frac    = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)     -- share of the kernel's wall cycles the matrix pipes run
    mfma_busy_frac_sq = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES * 32)            -- the same against the cycles an SQ had waves
The effective clock of the pass is GRBM_GUI_ACTIVE / 8 / kernel time (profiled passes clock lower than unprofiled ones)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

out, d = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row.get("Kernel_Name", "?").split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row.get("Kernel_Name", "?").split("(")[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:
    commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    commit = None
rec = {"commit": commit or os.environ.get("GIT_COMMIT"), "formulae": __doc__.split("Units")[1].strip(), "kernels": {}}
for k, ctrs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in ctrs.items()}
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or m.get("GRBM_GUI_ACTIVE", 0) <= 0:
        continue
    e = {"counters_mean": {c: round(v, 1) for c, v in sorted(m.items())}, "launches": len(next(iter(ctrs.values()))),
         "mfma_busy_frac": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)}
    if m.get("SQ_BUSY_CYCLES", 0) > 0:
        e["mfma_busy_frac_sq"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["SQ_BUSY_CYCLES"] * 32), 4)
    if m.get("SQ_INSTS_MFMA", 0) > 0:
        e["valu_per_mfma"] = round(m.get("SQ_ACTIVE_INST_VALU", 0) / 4 / m["SQ_INSTS_MFMA"], 2) if False else None
        e["lds_insts_per_mfma"] = round(m.get("SQ_INSTS_LDS", 0) / m["SQ_INSTS_MFMA"], 3)
    if dur.get(k):
        us = sum(dur[k]) / len(dur[k])
        e["us_per_launch_profiled"] = round(us, 1)
        e["effective_clock_ghz"] = round(m["GRBM_GUI_ACTIVE"] / 8 / us / 1e3, 3)
    e.pop("valu_per_mfma", None)
    rec["kernels"][k] = e
json.dump(rec, open(out, "w"), indent=1)
for k, e in rec["kernels"].items():
    print(f"{k[-70:]:<70s} mfma_busy_frac {e['mfma_busy_frac']:.3f}  (vs SQ busy {e.get('mfma_busy_frac_sq')})  clock {e.get('effective_clock_ghz')} GHz")
