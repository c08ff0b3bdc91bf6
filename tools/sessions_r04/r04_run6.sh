#!/bin/bash
# round 4, GPU session 6: kernel trace of ConsistI2V at the released model's width (16 f x 512^2 and 256^2), CLI test on the GPU
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT

for size in 512 256; do
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c2v_$size -o c2v -- python $GRAFT_REPO_ROOT/tools/consisti2v_bench.py $size 2 > $GRAFT_REPO_ROOT/gpurun_out/r04_c2v_prof_$size.log 2>&1)
python tools/summarize_profile.py gpurun_out/prof_c2v_$size --steps 8 > gpurun_out/r04_consisti2v_${size}_kernel_summary.md 2>&1
rm -rf gpurun_out/prof_c2v_$size
head -34 gpurun_out/r04_consisti2v_${size}_kernel_summary.md | cut -c1-150
done
