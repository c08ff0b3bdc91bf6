#!/bin/bash
# round 5: the whole -m gpu suite with the HIP error record (-> tests/golden/hip_error_caps.json), then smoke
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ANYV2V_RECORD_ERRS=gpurun_out/r05_hip_error_record.json timeout 2400 python -m pytest tests -m gpu -q -s --durations=12 > gpurun_out/r05_gputest_log_full.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r05_gputest_log_full.txt
grep -v "MIOpen\|it/s\]\|s/it\]\|amdgpu.ids" gpurun_out/r05_gputest_log_full.txt > gpurun_out/r05_gputest_log.txt
tail -25 gpurun_out/r05_gputest_log.txt | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.txt 2>&1; tail -3 gpurun_out/r05_smoke.txt | cut -c1-200
rm -f gpurun_out/r05_gputest_log_full.txt
