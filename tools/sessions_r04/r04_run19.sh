#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 900 python -m pytest tests/test_runners_e2e.py -q -x -m gpu -k "pipelined" > gpurun_out/r04_run19_tests.txt 2>&1; tail -3 gpurun_out/r04_run19_tests.txt | cut -c1-300
done
