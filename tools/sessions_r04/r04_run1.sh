#!/bin/bash
# round 4, GPU session 1: raster A/B (+FETCH_SIZE), config-5 parity rows with the chunked fp32 oracle, eager size probe
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "persistent_big_tile" > gpurun_out/r04_run1_gemm_big.log 2>&1; tail -3 gpurun_out/r04_run1_gemm_big.log
timeout 300 python tools/gemm_raster_ab.py > gpurun_out/r04_run1_raster.log 2>&1; tail -12 gpurun_out/r04_run1_raster.log
(cd /tmp && RASTER_PMC=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r04_raster_fetch -o t -- python $GRAFT_REPO_ROOT/tools/gemm_raster_ab.py > $GRAFT_REPO_ROOT/gpurun_out/r04_run1_raster_pmc.log 2>&1)
python tools/pmc_per_dispatch.py gemm_big gpurun_out/pmc_r04_raster_fetch > gpurun_out/r04_raster_fetch_per_dispatch.txt 2>&1; tail -50 gpurun_out/r04_raster_fetch_per_dispatch.txt
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -s -k "config5_midsize" > gpurun_out/r04_run1_midsize.log 2>&1; grep -E "^ok|^FAIL|passed|failed|diagnostic|B=" gpurun_out/r04_run1_midsize.log | cut -c1-420
ANYV2V_LONG_TESTS=1 ANYV2V_CONFIG5_B3_ONLY=1 ANYV2V_RECORD_ERRS=gpurun_out/r04_caps_config5.json timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "test_full_model_config5_step_vs_fp32_oracle" > gpurun_out/r04_run1_config5_full.log 2>&1; grep -E "^ok|^FAIL|passed|failed|diagnostic|B=" gpurun_out/r04_run1_config5_full.log | cut -c1-420
timeout 400 python tools/eager_size_probe.py > gpurun_out/r04_eager_size_probe_fp32.txt 2>&1; grep -E "DIFFERS|failed|done" gpurun_out/r04_eager_size_probe_fp32.txt | cut -c1-300
find gpurun_out -type f -size +6M -delete
