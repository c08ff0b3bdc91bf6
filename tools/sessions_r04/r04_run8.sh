#!/bin/bash
# round 4, GPU session 8: inversion / edit overlap experiment
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
python bench.py --steps 50 --warmup 2 --no-cpu-baseline --no-clip --no-multi-edit --no-roofline > gpurun_out/r04_overlap_off.log 2>&1; tail -1 gpurun_out/r04_overlap_off.log | cut -c1-400
python bench.py --steps 50 --warmup 2 --no-cpu-baseline --no-clip --no-multi-edit --no-roofline --overlap > gpurun_out/r04_overlap_on.log 2>&1; tail -1 gpurun_out/r04_overlap_on.log | cut -c1-300; grep -o '"overlap_bit_equal_to_serial": [a-z]*' gpurun_out/r04_overlap_on.log; tail -5 gpurun_out/r04_overlap_on.log | grep -i "error\|Traceback" 
