#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_runners_e2e.py -q -x -m gpu -k pipelined > gpurun_out/r04_run12_tests.txt 2>&1; tail -3 gpurun_out/r04_run12_tests.txt
timeout 900 python tools/job_pipeline_ab.py 6 50 > gpurun_out/r04_job_pipeline_ab.txt 2>&1; tail -1 gpurun_out/r04_job_pipeline_ab.txt
