#!/bin/bash
# round 4, last call: the ConsistI2V pipeline-level GPU checks after the sampler refactor (animation pipelines, guidance_rescale, eta)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ANYV2V_PRINT_ROWS=1 timeout 190 python -m pytest tests/test_gpu_parity.py -q -s -k "consisti2v_samplers or consisti2v_pipeline" > gpurun_out/r04_consisti2v_sampling_gpu_full.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r04_consisti2v_sampling_gpu_full.txt
grep -v "MIOpen\|it/s\]\|s/it\]\|amdgpu.ids" gpurun_out/r04_consisti2v_sampling_gpu_full.txt > gpurun_out/r04_consisti2v_sampling_gpu.txt
rm -f gpurun_out/r04_consisti2v_sampling_gpu_full.txt
tail -30 gpurun_out/r04_consisti2v_sampling_gpu.txt | cut -c1-220
