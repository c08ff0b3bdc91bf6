#!/bin/bash
# Round-6 session 2: parity of the one-wave-per-SIMD kernel, then its interleaved A/B against the default dispatch.
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "single_wave or persistent_big" > gpurun_out/r06_sw_parity.log 2>&1
tail -15 gpurun_out/r06_sw_parity.log
timeout 900 python tools/gemm_sw_ab.py > gpurun_out/r06_gemm_sw_ab.log 2>&1
tail -50 gpurun_out/r06_gemm_sw_ab.log
