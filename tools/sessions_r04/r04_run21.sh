#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python tools/job_pipeline_ab.py 3 500 > gpurun_out/r04_job_pipeline_ab_500.txt 2>&1; tail -1 gpurun_out/r04_job_pipeline_ab_500.txt
