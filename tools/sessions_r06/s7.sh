#!/bin/bash
# Round-6 session: parity of the LDS-patch 3x3 conv kernel (gemm_swh.hip), then its interleaved A/B.
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lds_patch" > gpurun_out/r06_halo_parity.log 2>&1
tail -25 gpurun_out/r06_halo_parity.log | cut -c1-300
timeout 900 python tools/gemm_sw_conv_order.py --product > gpurun_out/r06_conv_halo_ab.log 2>&1
tail -12 gpurun_out/r06_conv_halo_ab.log | cut -c1-400
