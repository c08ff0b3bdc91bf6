#!/bin/bash
# Round-6 session: parity of the stream-K form, then its interleaved A/B against the default dispatch (B = 1 and B = 3 shapes).
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "single_wave" > gpurun_out/r06_sk_parity.log 2>&1
tail -15 gpurun_out/r06_sk_parity.log
timeout 900 python tools/gemm_sw_ab.py --sk --tag r06_gemm_sk_ab > gpurun_out/r06_gemm_sk_ab.log 2>&1
tail -64 gpurun_out/r06_gemm_sk_ab.log
