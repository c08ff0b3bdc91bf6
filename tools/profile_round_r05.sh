#!/bin/bash
# Round-5 evidence session: default bench line, rocprofv3 kernel trace of the bench, FETCH / WRITE passes over the roofline kernels,
# whole-step traffic, per-shape reports.  Outputs under gpurun_out/ (copy what is judged into profiles/).
set -u
TAG=r05
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${TAG}_bench.log 2>&1
tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench_line.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-clip --no-multi-edit --no-configs --no-job-schedule > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1)
python tools/summarize_profile.py gpurun_out/prof_${TAG} --steps 22 > gpurun_out/${TAG}_bench_kernel_summary.md 2>&1
cp gpurun_out/prof_${TAG}/*/*kernel_stats.csv gpurun_out/${TAG}_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof_${TAG}
for grp in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${grp} -o t -- python $GRAFT_REPO_ROOT/tools/pmc_targets.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${grp}.log 2>&1)
done
python tools/pmc_traffic.py gpurun_out/${TAG}_traffic.json gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE > /dev/null 2>&1
python tools/pmc_per_dispatch.py gemm_big gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE > gpurun_out/${TAG}_pmc_gemm_big_per_dispatch.txt 2>&1
bash tools/step_traffic.sh ${TAG} > gpurun_out/${TAG}_step_traffic.log 2>&1
python tools/shape_report.py --batch 3 > gpurun_out/${TAG}_shape_report_B3.txt 2>&1
python tools/shape_report.py --batch 1 > gpurun_out/${TAG}_shape_report_B1.txt 2>&1
find gpurun_out -type f -size +6M -delete
tail -c 2500 gpurun_out/${TAG}_bench_line.json; echo; head -32 gpurun_out/${TAG}_bench_kernel_summary.md | cut -c1-140; tail -3 gpurun_out/${TAG}_step_traffic.log | cut -c1-600; head -12 gpurun_out/${TAG}_shape_report_B3.txt | cut -c1-150
