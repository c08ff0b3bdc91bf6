set -u
TAG=r05
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-clip --no-multi-edit --no-configs --no-job-schedule > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1)
python tools/summarize_profile.py gpurun_out/prof_${TAG} --steps 22 > gpurun_out/${TAG}_bench_kernel_summary.md 2>&1
cp gpurun_out/prof_${TAG}/*/*kernel_stats.csv gpurun_out/${TAG}_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof_${TAG}
head -30 gpurun_out/${TAG}_bench_kernel_summary.md | cut -c1-140
