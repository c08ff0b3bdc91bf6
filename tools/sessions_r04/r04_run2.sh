#!/bin/bash
# round 4, GPU session 2: ping-pong GEMM kernel -- parity, then A/B against the persistent kernel on the workload's shapes
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pingpong or persistent_big_tile" > gpurun_out/r04_run2_parity.log 2>&1; tail -25 gpurun_out/r04_run2_parity.log | cut -c1-300
export GEMM_AB_NO_PARITY=1
export GEMM_FORMS="0:big,131072:pp-auto,655360:pp192,1179648:pp256"
GEMM_CASES=conv GEMM_AB_OUT=r04_gemm_pp_ab_conv.txt timeout 400 python tools/gemm_epi_ab.py > gpurun_out/r04_run2_ab_conv.log 2>&1; tail -30 gpurun_out/r04_run2_ab_conv.log | cut -c1-400
GEMM_AB_OUT=r04_gemm_pp_ab_linear.txt timeout 400 python tools/gemm_epi_ab.py > gpurun_out/r04_run2_ab_lin.log 2>&1; tail -45 gpurun_out/r04_run2_ab_lin.log | cut -c1-400
