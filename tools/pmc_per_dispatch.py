"""Per-dispatch view of rocprofv3 --pmc counter CSVs (one line per kernel launch, in launch order), for passes where one kernel
name covers several shapes.  python tools/pmc_per_dispatch.py <name filter> <dir> [<dir> ...]"""
import collections
import csv
import glob
import os
import sys

flt = sys.argv[1]
for d in sys.argv[2:]:
    rows = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if flt not in row.get("Kernel_Name", ""):
                continue
            key = (int(row["Dispatch_Id"]), row["Kernel_Name"].split("(")[0][-48:], row.get("Grid_Size", ""))
            rows.setdefault(key, {})[row["Counter_Name"]] = float(row["Counter_Value"])
    print(f"# {d}")
    for (did, name, grid), ctr in sorted(rows.items()):
        print(f"  {did:5d} {name:<50s} " + "  ".join(f"{c}={v:.0f}" for c, v in sorted(ctr.items())))
