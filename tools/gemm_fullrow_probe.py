"""Experimental full-row epilogue of the persistent GEMM kernel (GEMM flag 64) vs the default wave-private epilogue:
bit-equality of the outputs and launch time on the K = 320 layers it targets.  Writes gpurun_out/gemm_fullrow_probe.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

dev = "cuda"
lines = []
ops.USE_GLDS = True


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def case(tag, M, N, K, mode=0, conv=None, res=False, rv=False):
    taps = 9 if mode == 1 else (3 if mode == 2 else 1)
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, taps * K, device=dev) / (taps * K) ** 0.5).half()
    b = torch.randn(N, device=dev).half()
    r = torch.randn(M, N, device=dev).half() if res else None
    rowvec = torch.randn(48, N, device=dev).half() if rv else None
    outs, ms = [], []
    for flags in (0, 64, 0, 64):
        ops.GEMM_FLAGS = flags
        out = torch.zeros(M, N, dtype=torch.float16, device=dev)
        fn = lambda: ops.gemm(a, w, bias=b, out=out, mode=mode, conv=conv, M=M, residual=r, rowvec=rowvec,
                              rowvec_div=M // 48 if rv else 0, temporal=(16, M // 48) if mode == 2 else None)
        ms.append(timeit(fn))
        outs.append(out.clone())
    ops.GEMM_FLAGS = 0
    same = bool(torch.equal(outs[0], outs[1])) and bool(torch.isfinite(outs[1].float()).all())
    lines.append(f"{tag:<40s} default {ms[0] * 1e3:7.1f} / {ms[2] * 1e3:7.1f} us   full-row {ms[1] * 1e3:7.1f} / {ms[3] * 1e3:7.1f} us   "
                 f"bit-equal {same}")
    print(lines[-1], flush=True)


T = 48 * 4096
case("warm-up", T, 320, 320, res=True)
case("linear 320->320 + res", T, 320, 320, res=True)
case("linear 320->960 (qkv)", T, 960, 320)
case("linear 1280->320 + res (ff down)", T, 320, 1280, res=True)
case("temporal conv 320 + res", T, 320, 320, mode=2, res=True)
case("conv3x3 320->320 + temb + res", T, 320, 320, mode=1, conv=(64, 64, 64, 64, 1, 0), res=True, rv=True)
case("linear 640->640 + res @32x32", T // 4, 640, 640, res=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_fullrow_probe.txt"), "w").write("\n".join(lines) + "\n")
