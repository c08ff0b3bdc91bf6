"""Per-shape cost table of one UNet forward (GPU): which (op, shape) pairs the step time is made of.

Records every ops.gemm / groupnorm / layernorm / attention call of one eager forward at the bench workload
(B=3 PnP edit step with all injections on, or B=1 inversion step), then re-times each unique shape in isolation with
HIP events.  Output: gpurun_out/shape_report_<B>.txt sorted by total time.

    python tools/shape_report.py [--batch 3|1] [--iters 5]
"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from anyv2v_amd import ops, pnp_utils  # noqa: E402
from anyv2v_amd.pipeline import I2VGenXLPipeline  # noqa: E402
from anyv2v_amd.schedulers import DDIMScheduler  # noqa: E402


def main():
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 3
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 5
    dev = torch.device("cuda", 0)
    torch.set_grad_enabled(False)
    pipe = I2VGenXLPipeline.from_pretrained("ali-vilab/i2vgen-xl", torch_dtype=torch.float16, variant="fp16", random_init_seed=0)
    pipe.to(dev)
    lat, ehs, ie, il = bench.synthetic_clip(dev, 8888)
    fwd = DDIMScheduler()
    fwd.set_timesteps(50)
    if B == 3:
        pnp_utils.register_conv_injection(pipe, fwd.timesteps)
        pnp_utils.register_spatial_attention_pnp(pipe, fwd.timesteps)
        pnp_utils.register_temp_attention_pnp(pipe, fwd.timesteps)
        pnp_utils.register_time(pipe, int(fwd.timesteps[0]))
        sample = lat.repeat(3, 1, 1, 1, 1).contiguous()
        cond = dict(encoder_hidden_states=ehs, fps=torch.tensor([8, 8, 8], device=dev), image_latents=il, image_embeddings=ie)
    else:
        pnp_utils.clear_time(pipe)
        sample = lat.clone()
        cond = dict(encoder_hidden_states=ehs[:1].contiguous(), fps=torch.tensor([8], device=dev),
                    image_latents=il[:1].contiguous(), image_embeddings=ie[:1].contiguous())
    t = int(fwd.timesteps[0])
    pipe.unet(sample, t, **cond)  # warm-up: packs weights, fills the per-clip cache
    torch.cuda.synchronize()

    calls = collections.OrderedDict()   # key -> [count, fn, args, kwargs, flops, bytes]

    def rec(name, orig, keyfn):
        def wrapped(*a, **k):
            key, flops, nbytes = keyfn(*a, **k)
            key = (name,) + key
            if key in calls:
                calls[key][0] += 1
            else:
                calls[key] = [1, orig, a, k, flops, nbytes]
            return orig(*a, **k)
        return wrapped

    def gemm_key(a0, w, *, a1=None, bias=None, rowvec=None, rowvec_div=0, residual=None, act=0, out=None, mode=0, conv=None,
                 temporal=None, M=None, naive=False, ln=None):
        M = M if M is not None else a0.shape[0]
        N, K = w.shape
        n_out = N // 2 if act == 3 else N
        nb = 2 * (a0.shape[0] * (a0.shape[1] + (a1.shape[1] if a1 is not None else 0)) + N * K + M * n_out * (2 if residual is not None else 1))
        return ((f"mode{mode}", f"act{act}", M, N, K, "2src" if a1 is not None else "", "res" if residual is not None else "",
                 "rv" if rowvec is not None else "", "ln" if ln is not None else "", str(conv or temporal or "")), 2.0 * M * N * K, nb)

    def gn_key(x0, gamma, beta, stats, rpg, *, x1=None, groups=32, eps=1e-5, silu=False, out=None, **_kw):
        C = x0.shape[1] + (x1.shape[1] if x1 is not None else 0)
        return ((x0.shape[0], C, rpg, "silu" if silu else "", "2src" if x1 is not None else ""), 0.0, 2 * x0.shape[0] * C * 3)

    def ln_key(x, gamma, beta, eps=1e-5, out=None):
        return ((x.shape[0], x.shape[1]), 0.0, 2 * x.shape[0] * x.shape[1] * 2)

    def attn_key(q, k, v, out, *, batch, heads, Sq, Sk, **kw):
        d = kw.get("head_dim", 64)
        return ((batch, heads, Sq, Sk, d, f"qk_mod{kw.get('qk_mod', 0)}", f"kv_div{kw.get('kv_div', 1)}"),
                4.0 * batch * heads * Sq * Sk * d, 2 * batch * heads * d * (2 * Sq + 2 * Sk / kw.get("kv_div", 1)))

    def ff_key(x, w1p, b1p, w2s, b2, residual=None, out=None):
        M, C = x.shape
        H = w2s.shape[0] * 32
        return ((M, C, H, "res" if residual is not None else ""), 2.0 * M * (2 * H * C + H * C),
                (M * C * (3 if residual is not None else 2) + 3 * H * C) * 2.0)

    saved_ff = ops.ff_geglu
    ops.ff_geglu = rec("ff_fused", ops.ff_geglu, ff_key)
    saved = (ops.gemm, ops.groupnorm, ops.layernorm, ops.attention)
    ops.gemm, ops.groupnorm = rec("gemm", ops.gemm, gemm_key), rec("groupnorm", ops.groupnorm, gn_key)
    ops.layernorm, ops.attention = rec("layernorm", ops.layernorm, ln_key), rec("attention", ops.attention, attn_key)
    pipe.unet(sample, t, **cond)
    ops.gemm, ops.groupnorm, ops.layernorm, ops.attention = saved
    ops.ff_geglu = saved_ff
    torch.cuda.synchronize()

    # whole-forward eager time for reference
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rows = []
    for key, (cnt, fn, a, k, flops, nb) in calls.items():
        fn(*a, **k)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        rows.append((cnt * us, cnt, us, flops / us / 1e6 if flops else 0.0, nb / us / 1e3, key))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    lines = [f"B={B}: {len(rows)} unique (op, shape) pairs, {sum(r[1] for r in rows)} calls, sum of isolated times {tot / 1e3:.2f} ms",
             f"{'total ms':>9s} {'%':>5s} {'cum%':>5s} {'n':>4s} {'us':>8s} {'TF/s':>7s} {'GB/s':>7s}  op / shape"]
    cum = 0.0
    for (tt, cnt, us, tf, gbs, key) in rows:
        cum += tt
        lines.append(f"{tt / 1e3:9.3f} {100 * tt / tot:5.1f} {100 * cum / tot:5.1f} {cnt:4d} {us:8.1f} {tf:7.1f} {gbs:7.0f}  "
                     + " ".join(str(x) for x in key if x != ""))
    by_op = collections.defaultdict(float)
    for r in rows:
        by_op[r[5][0] + (" " + r[5][1] + " " + r[5][2] if r[5][0] == "gemm" else "")] += r[0]
    lines.append("")
    for kname, v in sorted(by_op.items(), key=lambda kv: -kv[1]):
        lines.append(f"{v / 1e3:9.3f} ms {100 * v / tot:5.1f}%  {kname}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", f"shape_report_{B}.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main()
