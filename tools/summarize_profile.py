"""Summarise a rocprofv3 kernel trace (CSV) into the tables committed under profiles/.

    python tools/summarize_profile.py <dir-with-*_kernel_trace.csv> [--steps N] > profiles/rNN_<name>.md

Two tables: time per kernel name, and time per (kernel, grid) -- the latter separates e.g. the graded spatial
self-attention launches (7680 workgroups, S=4096) from cross-attention launches of the same kernel.
"""
import collections
import csv
import glob
import sys


def short(name: str) -> str:
    """Readable kernel label: strip 'void' / argument lists; demangle the few _Z names rocprofv3 leaves mangled
    (itanium: _Z<len><name>I<template args>E...) keeping integer / bool template arguments."""
    import re
    name = name.replace("void ", "")
    name = re.sub(r"\((GemmK|AttnK)[^)]*\)$", "", name)
    m = re.match(r"_Z(\d+)", name)
    if m:
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        rest = name[m.end() + n:]
        args = re.findall(r"L([ib])(\d+)E", rest.split("Ev")[0]) if rest.startswith("I") else []
        name = base + ("<" + ", ".join(("true" if v == "1" else "false") if t == "b" else v for t, v in args) + ">" if args else "")
    return name[:80]


def main():
    d = sys.argv[1]
    steps = None
    if "--steps" in sys.argv:
        steps = float(sys.argv[sys.argv.index("--steps") + 1])
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    assert files, "no *_kernel_trace.csv under " + d
    by_name = collections.defaultdict(lambda: [0, 0.0])
    by_grid = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            n = short(r["Kernel_Name"])
            by_name[n][0] += 1
            by_name[n][1] += us
            wg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) // max(1, int(r["Workgroup_Size_X"]))
            by_grid[(n, wg)].append(us)
    total = sum(v[1] for v in by_name.values())
    print(f"total kernel time {total / 1e3:.2f} ms over {sum(v[0] for v in by_name.values())} launches" +
          (f" ({total / 1e3 / steps:.2f} ms per bench step incl. warm-up/capture launches)" if steps else ""))
    print("\n| kernel | launches | total ms | % | avg us |\n|---|---:|---:|---:|---:|")
    ranked = sorted(by_name.items(), key=lambda kv: -kv[1][1])
    shown = [kv for kv in ranked if kv[1][1] >= 0.002 * total]          # every kernel with >= 0.2 % of the kernel time
    for n, (c, us) in shown:
        print(f"| `{n}` | {c} | {us / 1e3:.2f} | {100 * us / total:.1f} | {us / c:.1f} |")
    rest = ranked[len(shown):]
    if rest:
        print(f"| ({len(rest)} more kernels, each < 0.2 %) | {sum(v[0] for _, v in rest)} | {sum(v[1] for _, v in rest) / 1e3:.2f} | "
              f"{100 * sum(v[1] for _, v in rest) / total:.1f} | |")
    print("\n| kernel | workgroups | launches | avg us | min us | max us | total ms |\n|---|---:|---:|---:|---:|---:|---:|")
    for (n, wg), v in sorted(by_grid.items(), key=lambda kv: -sum(kv[1]))[:40]:
        print(f"| `{n}` | {wg} | {len(v)} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {sum(v) / 1e3:.2f} |")
    # the graded launch: plain spatial self-attention at (N=48, h=5, S=4096) = 7680 workgroups; cross-attention launches
    # (Sk = 145) share kernel and grid but run ~0.1 ms, so split on duration
    for (n, wg), v in by_grid.items():
        if n.startswith("flash_attn_d64_v2_kernel<3, 1, 8>") and wg == 3840:
            big = sorted(x for x in v if x > 600.0)
            if big:
                print(f"\nGraded launch (`{n}`, 3840 workgroups of 256 queries, self-attention N = 48, h = 5, S = Sk = 4096): "
                      f"{len(big)} launches, avg {sum(big) / len(big):.1f} us, median {big[len(big) // 2]:.1f} us, "
                      f"min {big[0]:.1f} us, max {big[-1]:.1f} us -- compare `roofline.ms_per_launch` of the same run's bench line.")


if __name__ == "__main__":
    main()
