#!/bin/bash
# Round-6 evidence session at HEAD: vendor kernel names, full -m gpu suite with the error record, SQ + FETCH / WRITE passes, step traffic,
# the default bench line, kernel trace summary, per-shape reports.
set -u
TAG=r06
export GIT_COMMIT=d3f1b22
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/vendor_trace -o v -- python $GRAFT_REPO_ROOT/tools/vendor_gemm_probe.py --vendor-only > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_vendor_trace.log 2>&1)
python tools/vendor_kernel_names.py gpurun_out/vendor_trace > gpurun_out/${TAG}_vendor_kernel_names.txt 2>&1
rm -rf gpurun_out/vendor_trace
ANYV2V_RECORD_ERRS=gpurun_out/${TAG}_hip_errors.json timeout 3000 python -m pytest tests -m gpu -x -q -rA > gpurun_out/${TAG}_gputest_log.txt 2>&1
tail -5 gpurun_out/${TAG}_gputest_log.txt
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_sq -o t -- python $GRAFT_REPO_ROOT/tools/pmc_targets.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_sq.log 2>&1)
python tools/pmc_sq.py gpurun_out/${TAG}_pmc.json gpurun_out/pmc_${TAG}_sq > gpurun_out/${TAG}_pmc_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_sq > gpurun_out/${TAG}_pmc_raw.txt 2>&1
rm -rf gpurun_out/pmc_${TAG}_sq
for grp in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${grp} -o t -- python $GRAFT_REPO_ROOT/tools/pmc_targets.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_${grp}.log 2>&1)
done
python tools/pmc_traffic.py gpurun_out/${TAG}_traffic.json gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE > /dev/null 2>&1
python tools/pmc_per_dispatch.py gemm_big gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE > gpurun_out/${TAG}_pmc_gemm_big_per_dispatch.txt 2>&1
rm -rf gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE
bash tools/step_traffic.sh ${TAG} > gpurun_out/${TAG}_step_traffic.log 2>&1
cp gpurun_out/${TAG}_pmc.json gpurun_out/${TAG}_traffic.json profiles/ 2>/dev/null
python bench.py > gpurun_out/${TAG}_bench.log 2>&1
tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench_line.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-clip --no-multi-edit --no-configs --no-job-schedule > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1)
python tools/summarize_profile.py gpurun_out/prof_${TAG} --steps 22 > gpurun_out/${TAG}_bench_kernel_summary.md 2>&1
cp gpurun_out/prof_${TAG}/*/*kernel_stats.csv gpurun_out/${TAG}_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof_${TAG}
python tools/shape_report.py --batch 3 > gpurun_out/${TAG}_shape_report_B3.txt 2>&1
python tools/shape_report.py --batch 1 > gpurun_out/${TAG}_shape_report_B1.txt 2>&1
find gpurun_out -type f -size +6M -delete
tail -c 1500 gpurun_out/${TAG}_bench_line.json; echo; head -12 gpurun_out/${TAG}_pmc_summary.txt; tail -3 gpurun_out/${TAG}_step_traffic.log | cut -c1-400
