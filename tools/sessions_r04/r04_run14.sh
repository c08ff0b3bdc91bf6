#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "released_width" > gpurun_out/r04_run14_tests.txt 2>&1; grep -v MIOpen gpurun_out/r04_run14_tests.txt | tail -8
