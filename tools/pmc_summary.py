"""Summarise rocprofv3 --pmc counter CSVs: mean of every counter per kernel name.
    python tools/pmc_summary.py <dir> [<dir> ...]   (searches for *counter_collection.csv)"""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "?")
            name = name.split("(")[0][-60:]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, ctrs in acc.items():
    print(f"## {name}")
    for c, v in sorted(ctrs.items()):
        print(f"  {c:<28s} mean {sum(v) / len(v):16.1f}   (n={len(v)})")
