#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or conv or norms or single_wave or lds_patch" > gpurun_out/r06_slice_major_parity.log 2>&1
tail -8 gpurun_out/r06_slice_major_parity.log | cut -c1-300
timeout 1500 python tools/conv_korder_ab.py > gpurun_out/r06_conv_slice_major_ab.log 2>&1
tail -22 gpurun_out/r06_conv_slice_major_ab.log | cut -c1-330
