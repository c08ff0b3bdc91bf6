#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "single_wave" > gpurun_out/r06_sw_parity2.log 2>&1
tail -4 gpurun_out/r06_sw_parity2.log | cut -c1-300
timeout 900 python tools/gemm_sw_ab.py --batches 3 --tag r06_gemm_sw_ab_gapless > gpurun_out/r06_gemm_sw_ab_gapless.log 2>&1
tail -30 gpurun_out/r06_gemm_sw_ab_gapless.log | cut -c1-200
