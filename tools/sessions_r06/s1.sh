#!/bin/bash
# Round-6 session 1: the vendor bar at HEAD (interleaved, every linear shape) + the vendor kernels' names, an SQ counter pass over
# the roofline launches (MFMA busy), and the bench line at HEAD on this box.
set -u
TAG=r06
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/vendor_gemm_probe.py > gpurun_out/${TAG}_vendor_probe.log 2>&1
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/vendor_trace -o v -- python $GRAFT_REPO_ROOT/tools/vendor_gemm_probe.py --vendor-only > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_vendor_trace.log 2>&1)
python tools/vendor_kernel_names.py gpurun_out/vendor_trace > gpurun_out/${TAG}_vendor_kernel_names.txt 2>&1
rm -rf gpurun_out/vendor_trace
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_sq -o t -- python $GRAFT_REPO_ROOT/tools/pmc_targets.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_sq.log 2>&1)
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_sq > gpurun_out/${TAG}_pmc_raw_head.txt 2>&1
rm -rf gpurun_out/pmc_${TAG}_sq
python bench.py --no-cpu-baseline --no-clip --no-multi-edit --no-job-schedule > gpurun_out/${TAG}_bench_s1.log 2>&1
tail -1 gpurun_out/${TAG}_bench_s1.log > gpurun_out/${TAG}_bench_line_s1.json
cat gpurun_out/${TAG}_vendor_gemm_probe.txt; head -60 gpurun_out/${TAG}_vendor_kernel_names.txt; tail -c 1500 gpurun_out/${TAG}_bench_line_s1.json
