#!/bin/bash
# round 4, GPU session 9: pipelined fused runner on the GPU (bit-equal files), default bench line with the pipelined schedule
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_runners_e2e.py -q -x -m gpu > gpurun_out/r04_run9_tests.txt 2>&1; tail -4 gpurun_out/r04_run9_tests.txt
python bench.py > gpurun_out/r04_bench_pipelined.log 2>&1; tail -1 gpurun_out/r04_bench_pipelined.log > gpurun_out/r04_bench_line_pipelined.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_line_pipelined.json"))
print({k: d[k] for k in ("value", "ms_per_step", "steps")}, d["config"].get("pipelined_bit_equal_to_serial"), d["config"].get("serial_ms_per_step"))
print(d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d.get("roofline_ff", {}).get("ms_per_launch"), d["flops"]["executed_tflop_per_step"])
PY
