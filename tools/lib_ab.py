"""A/B of two builds of the library on the default dispatch: tools/libanyv2v_hip_prev.so (the previous commit's gemm.hip) against the product
library, alternating PROCESSES on one box (each arm three times, median of medians), seeded inputs, and a checksum of every output so that
bit-equality between the two builds shows.  Shapes: the tile-kernel launches of the B = 3 edit step and the B = 1 inversion step.
    python tools/lib_ab.py [tag]      -> gpurun_out/<tag>.txt"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = []


def add(tag, M, N, K, mode=0, act=0, conv=None, temporal=None, res=False, rv=0, c1=0):
    CASES.append(dict(tag=tag, M=M, N=N, K=K, mode=mode, act=act, conv=conv, temporal=temporal, res=res, rv=rv, c1=c1))


for B, tb in ((3, "B3"), (1, "B1")):
    T0, T1, T2, T3 = B * 65536, B * 16384, B * 4096, B * 1024
    add(f"{tb} L0 conv3x3 +res", T0, 320, 2880, mode=1, conv=(64, 64, 64, 64, 1, 0), res=True)
    add(f"{tb} L0 conv3x3 +temb", T0, 320, 2880, mode=1, conv=(64, 64, 64, 64, 1, 0), rv=65536)
    add(f"{tb} L0 conv3x3 640->320 +temb", T0, 320, 5760, mode=1, conv=(64, 64, 64, 64, 1, 0), rv=65536, c1=320)
    add(f"{tb} L0 temporal conv", T0, 320, 960, mode=2, temporal=(16, 4096))
    add(f"{tb} L0 temporal conv +res", T0, 320, 960, mode=2, temporal=(16, 4096), res=True)
    add(f"{tb} L1 out-proj +res", T1, 640, 640, res=True)
    add(f"{tb} L1 QKV", T1, 1920, 640)
    add(f"{tb} L1 GEGLU", T1, 5120, 640, act=3)
    add(f"{tb} L1 FF down +res", T1, 640, 2560, res=True)
    add(f"{tb} L1 conv3x3 +res", T1, 640, 5760, mode=1, conv=(32, 32, 32, 32, 1, 0), res=True)
    add(f"{tb} L1 temporal conv", T1, 640, 1920, mode=2, temporal=(16, 1024))
    add(f"{tb} L2 out-proj +res", T2, 1280, 1280, res=True)
    add(f"{tb} L2 QKV", T2, 3840, 1280)
    add(f"{tb} L2 GEGLU", T2, 10240, 1280, act=3)
    add(f"{tb} L2 FF down +res", T2, 1280, 5120, res=True)
    add(f"{tb} L2 conv3x3 +res", T2, 1280, 11520, mode=1, conv=(16, 16, 16, 16, 1, 0), res=True)
    add(f"{tb} L2 temporal conv", T2, 1280, 3840, mode=2, temporal=(16, 256))
    add(f"{tb} L3 conv3x3 +res", T3, 1280, 11520, mode=1, conv=(8, 8, 8, 8, 1, 0), res=True)
    add(f"{tb} L3 GEGLU", T3, 10240, 1280, act=3)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from anyv2v_amd import _lib
    if os.environ.get("LIB_AB_PATH"):
        _lib.LIB_PATH = os.environ["LIB_AB_PATH"]
    from anyv2v_amd import ops
    for ci, c in enumerate(CASES):
        torch.manual_seed(1000 + ci)
        taps = {0: 1, 1: 9, 2: 3}[c["mode"]]
        cin = c["K"] // taps
        M, N = c["M"], c["N"]
        a = torch.randn(M, cin - c["c1"], device="cuda").half()
        a1 = torch.randn(M, c["c1"], device="cuda").half() if c["c1"] else None
        w = (torch.randn(N, c["K"], device="cuda") / c["K"] ** 0.5).half()
        b = torch.randn(N, device="cuda").half()
        n_out = N // 2 if c["act"] == 3 else N
        out = torch.empty(M, n_out, dtype=torch.float16, device="cuda")
        r = torch.randn(M, n_out, device="cuda").half() if c["res"] else None
        rowvec = torch.randn(M // c["rv"], N, device="cuda").half() if c["rv"] else None
        kw = dict(bias=b, mode=c["mode"], act=c["act"], conv=c["conv"], temporal=c["temporal"], residual=r, M=M, rowvec=rowvec, rowvec_div=c["rv"], a1=a1, out=out)
        for _ in range(3):
            ops.gemm(a, w, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm(a, w, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 4 * 1e3)
        h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"RESULT|{c['tag']}|{sorted(ts)[3]:.2f}|{h}", flush=True)
    sys.exit(0)

tag = sys.argv[1] if len(sys.argv) > 1 else "lib_ab"
arms = {"previous": os.path.join(ROOT, "tools", "libanyv2v_hip_prev.so"), "product": ""}
res = {a: {} for a in arms}
chk = {a: {} for a in arms}
for rep in range(3):
    for a, lib in arms.items():
        out = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LIB_AB_PATH=lib), capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith("RESULT|"):
                _, t, us, h = line.split("|")
                res[a].setdefault(t, []).append(float(us))
                chk[a][t] = h
lines, tot = [], {a: 0.0 for a in arms}
for c in CASES:
    t = {a: sorted(res[a][c["tag"]])[len(res[a][c["tag"]]) // 2] for a in arms}
    for a in arms:
        tot[a] += t[a]
    fl = 2.0 * c["M"] * c["N"] * c["K"]
    lines.append(f"{c['tag']:<32s}: previous {t['previous']:8.1f} us ({fl / t['previous'] / 1e6:5.0f} TF) | product {t['product']:8.1f} us ({fl / t['product'] / 1e6:5.0f} TF) | "
                 f"ratio {t['product'] / t['previous']:5.3f} | bit-equal {chk['previous'][c['tag']] == chk['product'][c['tag']]}")
    print(lines[-1], flush=True)
lines.append(f"sum: previous {tot['previous']:.1f} us -> product {tot['product']:.1f} us ({tot['product'] / tot['previous']:.4f})")
print(lines[-1])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", tag + ".txt"), "w").write("\n".join(lines) + "\n")
