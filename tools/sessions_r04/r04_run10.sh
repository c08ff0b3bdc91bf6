#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-clip --no-multi-edit > gpurun_out/r04_bench_pipelined2.log 2>&1; tail -1 gpurun_out/r04_bench_pipelined2.log > gpurun_out/r04_bench_line_pipelined2.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_line_pipelined2.json"))
print({k: d[k] for k in ("value", "ms_per_step", "steps")}, d["config"].get("pipelined_bit_equal_to_serial"), d["config"].get("serial_ms_per_step"))
PY
tail -3 gpurun_out/r04_bench_pipelined2.log | cut -c1-300
