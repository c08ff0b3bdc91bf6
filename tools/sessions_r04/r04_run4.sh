#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pingpong" > gpurun_out/r04_run4_parity.log 2>&1; tail -3 gpurun_out/r04_run4_parity.log | cut -c1-300
export GEMM_AB_NO_PARITY=1
export GEMM_FORMS="0:base,131072:pp-auto,655360:pp192,1179648:pp256"
GEMM_CASES=conv GEMM_AB_OUT=r04_gemm_pp_ab_b1_conv.txt timeout 400 python tools/gemm_epi_ab.py 2>&1 | grep "^B1" | cut -c1-330
GEMM_AB_OUT=r04_gemm_pp_ab_b1_linear.txt timeout 400 python tools/gemm_epi_ab.py 2>&1 | grep "^B1" | cut -c1-330
