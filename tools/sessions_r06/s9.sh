#!/bin/bash
# Round-6: the opt-in 500-step inversion drift row at HEAD (last run: round 3), smoke, and one more default bench line on another box.
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
ANYV2V_LONG_TESTS=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -rA -k "500 or long or drift" > gpurun_out/r06_gputest_long_log.txt 2>&1
tail -6 gpurun_out/r06_gputest_long_log.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.txt 2>&1; tail -3 gpurun_out/r06_smoke.txt
python bench.py > gpurun_out/r06_bench_b.log 2>&1
tail -1 gpurun_out/r06_bench_b.log > gpurun_out/r06_bench_line_box_b.json
python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_line_box_b.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['configs']['config_2_inversion_only']['frames_per_s'])"
