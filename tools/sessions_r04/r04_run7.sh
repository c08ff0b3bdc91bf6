#!/bin/bash
# round 4, GPU session 7: small attention at head_dim 40 / 160, ConsistI2V re-timed
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -s -k "small_mfma or consisti2v or clip" > gpurun_out/r04_run7_tests.txt 2>&1; tail -5 gpurun_out/r04_run7_tests.txt
timeout 600 python tools/consisti2v_bench.py 256 4 > gpurun_out/r04_consisti2v_256.txt 2>&1; tail -1 gpurun_out/r04_consisti2v_256.txt
timeout 600 python tools/consisti2v_bench.py 512 4 > gpurun_out/r04_consisti2v_512.txt 2>&1; tail -1 gpurun_out/r04_consisti2v_512.txt
for size in 512; do
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c2v_$size -o c2v -- python $GRAFT_REPO_ROOT/tools/consisti2v_bench.py $size 2 > $GRAFT_REPO_ROOT/gpurun_out/r04_c2v_prof_$size.log 2>&1)
python tools/summarize_profile.py gpurun_out/prof_c2v_$size --steps 8 > gpurun_out/r04_consisti2v_${size}_kernel_summary.md 2>&1
rm -rf gpurun_out/prof_c2v_$size
head -30 gpurun_out/r04_consisti2v_${size}_kernel_summary.md | cut -c1-150
done
