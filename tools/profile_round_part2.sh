set -u
TAG=r03
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -o bench -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-clip --no-multi-edit > gpurun_out/${TAG}_prof_bench.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_${TAG} --steps 22 > gpurun_out/${TAG}_bench_kernel_summary.md 2>&1
cp gpurun_out/prof_${TAG}/*/*kernel_stats.csv gpurun_out/${TAG}_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof_${TAG}
for grp in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_${TAG}_${grp} -o t -- python tools/pmc_targets.py > gpurun_out/pmc_${TAG}_${grp}.log 2>&1
done
python tools/pmc_traffic.py gpurun_out/${TAG}_traffic.json gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE > /dev/null 2>&1
bash tools/step_traffic.sh r03 > gpurun_out/r03_step_traffic.log 2>&1
python tools/shape_report.py --batch 3 > gpurun_out/r03_shape3.log 2>&1
python tools/shape_report.py --batch 1 > gpurun_out/r03_shape1.log 2>&1
find gpurun_out -type f -size +6M -delete
du -sh gpurun_out; tail -3 gpurun_out/r03_step_traffic.log; head -12 gpurun_out/r03_bench_kernel_summary.md
