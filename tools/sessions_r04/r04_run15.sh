#!/bin/bash
set -u
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_consisti2v.py -q -x -m gpu -k "consisti2v" > gpurun_out/r04_run15_tests.txt 2>&1; grep -v MIOpen gpurun_out/r04_run15_tests.txt | tail -4
timeout 600 python tools/consisti2v_bench.py 256 10 > gpurun_out/r04_consisti2v_256.txt 2>&1; tail -1 gpurun_out/r04_consisti2v_256.txt
timeout 600 python tools/consisti2v_bench.py 512 10 > gpurun_out/r04_consisti2v_512.txt 2>&1; tail -1 gpurun_out/r04_consisti2v_512.txt
ANYV2V_NO_GRAPH=1 timeout 600 python tools/consisti2v_bench.py 256 10 > gpurun_out/r04_consisti2v_256_eager.txt 2>&1; tail -1 gpurun_out/r04_consisti2v_256_eager.txt
