"""The vendor bar for the hand-written GEMM kernels: torch.matmul (hipBLASLt / rocBLAS, fp16 in, fp32 accumulate) against ops.gemm
on every plain-linear (op, shape) of the step pair (profiles/r05_shape_report_B{1,3}.txt), random fp16 operands, the two arms
INTERLEAVED launch by launch on one box.  The product never calls the vendor library; this is the measuring stick.
    python tools/vendor_gemm_probe.py                 -> gpurun_out/r06_vendor_gemm_probe.txt
    python tools/vendor_gemm_probe.py --vendor-only   (3 vendor launches per shape, nothing else: run under
        rocprofv3 --kernel-trace to read the Tensile kernel name -- macro tile, MFMA shape, K depth -- per shape;
        tools/vendor_kernel_names.py turns the trace into a table)
GEGLU rows time OUR fused launch (matmul + bias + h * gelu(gate), half-width output) against the vendor's bare matmul of the
same M x N x K: the vendor writes twice the bytes and does no activation."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import ops  # noqa: E402

# (M, N, K, kind)  kind: "" plain, "res" +residual, "geglu", "ln" (LayerNorm-folded QKV: timed here as the plain launch)
SHAPES = [
    # B = 3 (edit step)
    (196608, 320, 320, "res"), (196608, 320, 320, ""), (196608, 960, 320, ""), (196608, 1536, 512, ""),
    (196608, 512, 2048, "res"), (196608, 4096, 512, "geglu"),
    (49152, 640, 640, "res"), (49152, 640, 640, ""), (49152, 1920, 640, ""), (49152, 640, 2560, "res"),
    (49152, 5120, 640, "geglu"),
    (12288, 1280, 1280, "res"), (12288, 1280, 1280, ""), (12288, 3840, 1280, ""), (12288, 1280, 5120, "res"),
    (12288, 10240, 1280, "geglu"),
    (3072, 1280, 1280, "res"), (3072, 3840, 1280, ""), (3072, 1280, 5120, "res"), (3072, 10240, 1280, "geglu"),
    # B = 1 (inversion step)
    (65536, 320, 320, "res"), (65536, 960, 320, ""), (65536, 1536, 512, ""), (65536, 4096, 512, "geglu"),
    (16384, 640, 640, "res"), (16384, 1920, 640, ""), (16384, 640, 2560, "res"), (16384, 5120, 640, "geglu"),
    (4096, 1280, 1280, "res"), (4096, 3840, 1280, ""), (4096, 1280, 5120, "res"), (4096, 10240, 1280, "geglu"),
    (1024, 1280, 1280, "res"), (1024, 3840, 1280, ""), (1024, 1280, 5120, "res"), (1024, 10240, 1280, "geglu"),
    # the long-K shapes of the round-1 probe (implicit-GEMM sizes of the 16x16-level convolutions as plain matmuls)
    (12288, 1280, 11520, ""), (49152, 640, 5760, ""), (196608, 320, 2880, ""),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vendor-only", action="store_true")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_vendor_gemm_probe.txt"))
    args = ap.parse_args()
    dev = "cuda"
    lines = []
    for (M, N, K, kind) in SHAPES:
        a = torch.randn(M, K, device=dev).half()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
        geglu = kind == "geglu"
        out = torch.empty(M, N // 2 if geglu else N, dtype=torch.float16, device=dev)
        vout = torch.empty(M, N, dtype=torch.float16, device=dev)
        bias = torch.zeros(N, dtype=torch.float16, device=dev)
        res = torch.randn(M, N, device=dev).half() if kind == "res" else None

        def mine():
            if geglu:
                ops.gemm(a, w, bias=bias, act=ops.ACT_GEGLU, out=out)
            elif res is not None:
                ops.gemm(a, w, bias=bias, residual=res, out=out)
            else:
                ops.gemm(a, w, bias=bias, out=out)

        def vendor():
            torch.matmul(a, w.t(), out=vout)

        if args.vendor_only:
            for _ in range(3):
                vendor()
            torch.cuda.synchronize()
            continue
        for _ in range(3):
            mine()
            vendor()
        torch.cuda.synchronize()
        tm, tv = [], []
        for _ in range(args.rounds):
            for fn, acc in ((mine, tm), (vendor, tv)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                acc.append(e0.elapsed_time(e1) / 4 * 1e3)
        tm.sort()
        tv.sort()
        t_mine, t_v = tm[len(tm) // 2], tv[len(tv) // 2]
        fl = 2.0 * M * N * K
        lines.append(f"M={M:6d} N={N:5d} K={K:5d} {kind:5s}: ours {t_mine:7.1f} us ({fl / t_mine / 1e6:6.0f} TF) | vendor A@W^T {t_v:7.1f} us "
                     f"({fl / t_v / 1e6:6.0f} TF) | ours/vendor time {t_mine / t_v:5.2f}")
        print(lines[-1], flush=True)
        del a, w, out, vout, res
    if not args.vendor_only:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        open(args.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
