#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_runners_e2e.py tests/test_consisti2v.py tests/test_seine.py -q -x -m gpu > gpurun_out/r04_run23_a.txt 2>&1; tail -2 gpurun_out/r04_run23_a.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "pipeline_loops or consisti2v_pipeline or vs_the_reference_pipelines_own_output or elementwise" > gpurun_out/r04_run23_b.txt 2>&1; tail -2 gpurun_out/r04_run23_b.txt
