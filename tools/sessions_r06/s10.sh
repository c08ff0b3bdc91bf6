#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(cd /tmp && rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_l2 -o t -- python $GRAFT_REPO_ROOT/tools/l2_hit_targets.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_l2.log 2>&1)
python - <<'PY'
import csv, glob, collections
rows=[]
for f in glob.glob("gpurun_out/pmc_l2/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
by=collections.OrderedDict()
for r in rows:
    k=(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-60:])
    by.setdefault(k,{})[r["Counter_Name"]]=float(r["Counter_Value"])
out=[]
for (d,k),c in by.items():
    if "gemm" not in k: continue
    h,m,q=c.get("TCC_HIT_sum",0),c.get("TCC_MISS_sum",0),c.get("TCC_REQ_sum",0)
    out.append(f"{d:>5s} {k:<62s} req {q:14.0f} hit {h:14.0f} miss {m:14.0f} hit-rate {h/max(1,h+m):.3f}")
open("gpurun_out/r06_l2_hit_rates.txt","w").write("\n".join(out)+"\n")
print("\n".join(out))
PY
rm -rf gpurun_out/pmc_l2
