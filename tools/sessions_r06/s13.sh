#!/bin/bash
# Round-6 final validation at HEAD: whole -m gpu suite, smoke, default bench line.
set -u
export GIT_COMMIT=1e57236
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest_final_log.txt 2>&1
tail -4 gpurun_out/r06_gputest_final_log.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke_final.txt 2>&1; tail -2 gpurun_out/r06_smoke_final.txt | cut -c1-200
python bench.py > gpurun_out/r06_bench_final.log 2>&1
tail -1 gpurun_out/r06_bench_final.log > gpurun_out/r06_bench_line_final.json
python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_line_final.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('mfma_busy_frac'), d['configs']['config_2_inversion_only']['frames_per_s'], d['step_traffic']['hbm_gb_per_step_pair'], d['step_traffic'].get('commit'))"
