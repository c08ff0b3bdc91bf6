#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
ANYV2V_PRINT_ROWS=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_frame_parallel.py -q -s -m gpu -k "vae or frame_parallel or consisti2v or seine or small_mfma or bias" > gpurun_out/r04_rows.txt 2>&1
grep "^ok \|^FAIL\|passed\|failed" gpurun_out/r04_rows.txt | grep -i "vae\|frame-parallel\|frame parallel\|rank \|consisti2v\|seine\|passed\|failed" | cut -c1-210
