#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for f in 1 0 1; do
ANYV2V_PIPELINE_PACING=$f python bench.py --no-cpu-baseline --no-clip --no-multi-edit --no-roofline > gpurun_out/r04_pacing_$f.log 2>&1
python - <<PY
import json
d = json.loads(open("gpurun_out/r04_pacing_$f.log").read().strip().splitlines()[-1])
print("pacing $f", d["ms_per_step"], d["config"].get("serial_ms_per_step"), d["config"].get("pipelined_bit_equal_to_serial"))
PY
done
