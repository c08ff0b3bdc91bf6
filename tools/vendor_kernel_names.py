"""Turn a `rocprofv3 --kernel-trace` CSV of `tools/vendor_gemm_probe.py --vendor-only` into a per-shape table of the vendor
kernel that ran (Tensile name: macro tile MT, MFMA shape MI, depth-U, wave layout WG, LDS / prefetch options) and its duration.
    python tools/vendor_kernel_names.py <trace dir>"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vendor_gemm_probe import SHAPES  # noqa: E402

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def is_torch_helper(n):
    return n.startswith("void at::") or n.startswith("at::") or "elementwise" in n or "rocclr" in n or "distribution" in n
gemms = [r for r in rows if not is_torch_helper(r["Kernel_Name"])]
import collections
print(f"{len(rows)} dispatches, {len(gemms)} vendor GEMM dispatches, {len(SHAPES)} shapes x 3")
for n, c in collections.Counter(r["Kernel_Name"][:60] for r in gemms).most_common():
    print(f"   {c:4d} x {n}")
per = len(gemms) // max(1, len(SHAPES))
for i, (M, N, K, kind) in enumerate(SHAPES):
    grp = gemms[i * per:(i + 1) * per]
    if not grp:
        continue
    last = grp[-1]
    us = (int(last["End_Timestamp"]) - int(last["Start_Timestamp"])) / 1e3
    names = sorted({g["Kernel_Name"] for g in grp})
    print(f"M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF  grid {last.get('Grid_Size_X', '?')} wg {last.get('Workgroup_Size_X', '?')} "
          f"lds {last.get('LDS_Block_Size', '?')} vgpr {last.get('VGPR_Count', '?')} agpr {last.get('Accum_VGPR_Count', '?')}")
    for n in names:
        import re
        key = re.findall(r"(MT\d+x\d+x\d+|MI\d+x\d+x\d+|MIWT\d+_\d+|PGR\d|PLR\d|SK\d|WG\d+_\d+_\d+|LDSB\d|DTLA\d|DTLB\d|DTVA\d|DTVB\d|1LDSB\d|GRVWA\d+|GRVWB\d+|LRVW\d+|TLDS\d|SU\d+|SUM\d|SUS\d+)", n)
        print(f"      {' '.join(key) if key else n[:120]}")
