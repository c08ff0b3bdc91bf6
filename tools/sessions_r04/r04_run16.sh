#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_seine.py -q -x -m gpu -k "seine or small_mfma or bias" > gpurun_out/r04_run16_tests.txt 2>&1; grep -v "MIOpen\|it/s\|s/it" gpurun_out/r04_run16_tests.txt | tail -12
timeout 600 python tools/seine_bench.py 320 512 > gpurun_out/r04_seine_320x512.txt 2>&1; tail -1 gpurun_out/r04_seine_320x512.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_seine -o s -- python $GRAFT_REPO_ROOT/tools/seine_bench.py 320 512 > $GRAFT_REPO_ROOT/gpurun_out/r04_seine_prof.log 2>&1)
python tools/summarize_profile.py gpurun_out/prof_seine --steps 8 > gpurun_out/r04_seine_kernel_summary.md 2>&1
rm -rf gpurun_out/prof_seine
head -14 gpurun_out/r04_seine_kernel_summary.md | cut -c1-140
