"""A/B of the conv2d K order on the default dispatch: tools/libanyv2v_hip_prev.so (tap-major (tap, slice) gather, the library of the commit
before) against the product library (slice-major (slice, tap) with the incremental per-row addressing), alternating PROCESSES on one box
(each arm three times, median of medians), every 3x3 conv shape of the step pair incl. stride 2, the folded nearest-x2 up-sampler and the
two-source (skip concat) convs.  gpurun_out/r06_conv_slice_major_ab.txt
The slice-major form is tools/experiments/conv_slice_major_default_path.patch (v2: pointer math at issue time; v1 had it in next());
build the product library with the patch applied and the un-patched gemm.hip / gemm_sw.hip into tools/libanyv2v_hip_prev.so."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [  # tag, images, Hi, Ho, stride, up, c0, c1, cout, res, temb
    ("B3 64x64 320->320 +temb", 48, 64, 64, 1, 0, 320, 0, 320, 0, 1), ("B3 64x64 320->320 +res", 48, 64, 64, 1, 0, 320, 0, 320, 1, 0),
    ("B3 64x64 640->320 (320+320)", 48, 64, 64, 1, 0, 320, 320, 320, 0, 1), ("B3 64x64 960->320 (640+320)", 48, 64, 64, 1, 0, 640, 320, 320, 0, 1),
    ("B3 64->32 stride 2 320->320", 48, 64, 32, 2, 0, 320, 0, 320, 0, 0), ("B3 32x32 320->640 +temb", 48, 32, 32, 1, 0, 320, 0, 640, 0, 1),
    ("B3 32x32 640->640 +res", 48, 32, 32, 1, 0, 640, 0, 640, 1, 0), ("B3 32x32 1280->640 (640+640)", 48, 32, 32, 1, 0, 640, 640, 640, 0, 1),
    ("B3 32->64 nearest x2 640->640", 48, 32, 64, 1, 1, 640, 0, 640, 0, 0), ("B3 16x16 1280->1280 +res", 48, 16, 16, 1, 0, 1280, 0, 1280, 1, 0),
    ("B3 16x16 2560->1280 (1280+1280)", 48, 16, 16, 1, 0, 1280, 1280, 1280, 0, 1), ("B3 16->32 nearest x2 1280->1280", 48, 16, 32, 1, 1, 1280, 0, 1280, 0, 0),
    ("B3 8x8 1280->1280 +res", 48, 8, 8, 1, 0, 1280, 0, 1280, 1, 0), ("B3 8x8 2560->1280 (1280+1280)", 48, 8, 8, 1, 0, 1280, 1280, 1280, 0, 1),
    ("B1 64x64 320->320 +res", 16, 64, 64, 1, 0, 320, 0, 320, 1, 0), ("B1 32x32 640->640 +res", 16, 32, 32, 1, 0, 640, 0, 640, 1, 0),
    ("B1 16x16 1280->1280 +res", 16, 16, 16, 1, 0, 1280, 0, 1280, 1, 0), ("B1 8x8 1280->1280 +res", 16, 8, 8, 1, 0, 1280, 0, 1280, 1, 0),
]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from anyv2v_amd import _lib
    if os.environ.get("CONV_AB_LIB"):
        _lib.LIB_PATH = os.environ["CONV_AB_LIB"]
    from anyv2v_amd import ops
    for (tag, n, Hi, Ho, stride, up, c0, c1, co, res, temb) in CASES:
        M = n * Ho * Ho
        x0 = torch.randn(n * Hi * Hi, c0, device="cuda").half()
        x1 = torch.randn(n * Hi * Hi, c1, device="cuda").half() if c1 else None
        K = 9 * (c0 + c1)
        w = (torch.randn(co, K, device="cuda") / K ** 0.5).half()
        b = torch.randn(co, device="cuda").half()
        r = torch.randn(M, co, device="cuda").half() if res else None
        rv = torch.randn(n // 16, co, device="cuda").half() if temb else None
        out = torch.empty(M, co, dtype=torch.float16, device="cuda")
        kw = dict(a1=x1, bias=b, residual=r, rowvec=rv, rowvec_div=16 * Ho * Ho if temb else 0, mode=ops.MODE_CONV2D, conv=(Hi, Hi, Ho, Ho, stride, up), M=M, out=out)
        for _ in range(3):
            ops.gemm(x0, w, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm(x0, w, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 4 * 1e3)
        print(f"RESULT|{tag}|{sorted(ts)[3]:.2f}|{float(out.float().abs().mean()):.6f}", flush=True)
    sys.exit(0)
arms = {"tap-major (previous library)": os.path.join(ROOT, "tools", "libanyv2v_hip_prev.so"), "slice-major (product)": ""}
res = {a: {} for a in arms}
chk = {a: {} for a in arms}
for rep in range(3):
    for a, lib in arms.items():
        env = dict(os.environ, CONV_AB_LIB=lib)
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith("RESULT|"):
                _, tag, us, c = line.split("|")
                res[a].setdefault(tag, []).append(float(us))
                chk[a][tag] = float(c)
lines = []
tot = {a: 0.0 for a in arms}
for (tag, n, Hi, Ho, stride, up, c0, c1, co, _, _) in CASES:
    fl = 2.0 * n * Ho * Ho * co * 9 * (c0 + c1)
    t = {a: sorted(res[a][tag])[len(res[a][tag]) // 2] for a in arms}
    a0, a1 = list(arms)
    for a in arms:
        tot[a] += t[a]
    lines.append(f"{tag:<34s}: {a0} {t[a0]:8.1f} us ({fl / t[a0] / 1e6:5.0f} TF) | {a1} {t[a1]:8.1f} us ({fl / t[a1] / 1e6:5.0f} TF) | ratio {t[a1] / t[a0]:5.3f} | "
                 f"mean|out| {chk[a0][tag]:.5f} / {chk[a1][tag]:.5f}")
    print(lines[-1], flush=True)
lines.append(f"sum: {tot[list(arms)[0]]:.1f} us -> {tot[list(arms)[1]]:.1f} us")
print(lines[-1])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r06_conv_slice_major_ab.txt"), "w").write("\n".join(lines) + "\n")
