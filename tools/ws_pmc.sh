set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/wspmc_${name} -o t -- python tools/ws_pmc_targets.py 2 > gpurun_out/wspmc_${name}.log 2>&1
done
python tools/pmc_per_dispatch.py gemm_ws gpurun_out/wspmc_* > gpurun_out/ws_pmc_v2.txt 2>&1
python - <<'PY' >> gpurun_out/ws_pmc_v2.txt 2>&1
import csv, glob
for d in sorted(glob.glob("gpurun_out/wspmc_GRBM_GUI_ACTIVE")):
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_ws" in r["Kernel_Name"]:
                print("dur", r["Dispatch_Id"] if "Dispatch_Id" in r else "", r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
rm -rf gpurun_out/wspmc_*/*/*.db
cat gpurun_out/ws_pmc_v2.txt
