#!/bin/bash
# round 4, GPU session 5: ConsistI2V UNet + pipeline parity on the kernels, full-width step timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -x -k "consisti2v" > gpurun_out/r04_run5_tests.txt 2>&1
echo "tests exit $?" >> gpurun_out/r04_run5_tests.txt
tail -30 gpurun_out/r04_run5_tests.txt
timeout 600 python tools/consisti2v_bench.py 256 4 > gpurun_out/r04_consisti2v_256.txt 2>&1; tail -3 gpurun_out/r04_consisti2v_256.txt
timeout 600 python tools/consisti2v_bench.py 512 2 > gpurun_out/r04_consisti2v_512.txt 2>&1; tail -3 gpurun_out/r04_consisti2v_512.txt
