#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python tools/batched_inversion_probe.py 2 4 > gpurun_out/r04_batched_inversion_probe.txt 2>&1; tail -1 gpurun_out/r04_batched_inversion_probe.txt
timeout 600 python -m pytest tests/test_runners_e2e.py -q -x -m gpu -k "batched_inversion" > gpurun_out/r04_run22_tests.txt 2>&1; tail -2 gpurun_out/r04_run22_tests.txt
