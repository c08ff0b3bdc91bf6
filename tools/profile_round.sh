#!/bin/bash
# One GPU session: default bench line, rocprofv3 kernel trace of the bench, PMC passes (one counter group each) over the
# roofline kernels.  Outputs under gpurun_out/ (copy what is judged into profiles/).   bash tools/profile_round.sh <tag>
set -u
TAG=${1:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py > gpurun_out/${TAG}_bench.log 2>&1
tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -o bench -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-clip --no-multi-edit > gpurun_out/${TAG}_prof_bench.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_${TAG} --steps 22 > gpurun_out/${TAG}_bench_kernel_summary.md 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_${TAG}_${name} -o t -- python tools/pmc_targets.py > gpurun_out/pmc_${TAG}_${name}.log 2>&1
done
python tools/pmc_traffic.py gpurun_out/${TAG}_traffic.json gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_* > gpurun_out/${TAG}_pmc_raw.txt 2>&1
# kernel durations inside the PMC passes' traces (for the clock: GRBM_GUI_ACTIVE / duration)
python - <<'PY' > gpurun_out/${TAG}_pmc_durations.txt 2>&1
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_*_GRBM_GUI_ACTIVE")) + sorted(glob.glob("gpurun_out/pmc_*_FETCH_SIZE")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][-70:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(d)
    for k, v in acc.items():
        print(f"  {k:<72s} n={len(v)} mean {sum(v)/len(v):9.1f} us  min {min(v):9.1f}")
PY
rm -rf gpurun_out/prof_${TAG}/*/*.db 2>/dev/null
tail -c 3000 gpurun_out/${TAG}_bench_line.json
echo
head -30 gpurun_out/${TAG}_bench_kernel_summary.md
cat gpurun_out/${TAG}_pmc_durations.txt
# the merge back from the GPU box is capped at 64 MiB: keep the summaries and the stats CSVs, drop the raw per-dispatch traces
mkdir -p gpurun_out/${TAG}_keep
cp gpurun_out/prof_${TAG}/*/*kernel_stats.csv gpurun_out/${TAG}_bench_kernel_stats.csv 2>/dev/null
du -sh gpurun_out/* 2>/dev/null | sort -h | tail -8
find gpurun_out -type f -size +6M -delete
rmdir gpurun_out/${TAG}_keep 2>/dev/null
