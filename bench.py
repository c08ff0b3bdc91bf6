"""bench.py -- frames/sec for 16f x 512x512 DDIM-inversion + PnP-edit on MI355X (BASELINE.json metric).

A "step" here is ONE inversion denoise step (UNet B=1 + inverse-DDIM update + trajectory write) PLUS ONE PnP-edit
denoise step (source-latent read, UNet B=3 with conv / spatial / temporal feature injection, CFG 9.0 + DDIM update)
on one synthetic 16-frame 512x512 clip, i.e. 1/50 of BASELINE config 3 (50-step inversion + 50-step edit);
--steps 50 is exactly one clip.  frames/sec = 16 * (K / 50) * n_gpus / seconds.
The timed region is the SERIAL single-clip order (inversion step, then edit step, one stream): BASELINE config 3 is ONE clip, and
SURVEY 8(d) defines frames/s as 16 / wall-seconds for one clip, so `value` = 16 / 50 / ms_per_step.  The job-level two-clip software
pipeline (next clip's inversion beside this clip's edit; run_group_anyv2v) is measured after the timed region and reported as
`config.pipelined_ms_per_step` / `config.job_frames_per_s` (`--pipelined` times it instead).  `configs` adds BASELINE config 2
(inversion alone) and config 5 (128 frames: one B=1 and one B=3 step at full size).
Everything inside the step runs in the hand-written HIP kernels (anyv2v_amd/libanyv2v_hip.so); weights are random
(seeded) with the exact I2VGen-XL architecture, inputs synthetic and HBM-resident; VAE/CLIP pre/post are outside.

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STEPS_PER_STAGE = 50          # BASELINE config 2/3: 50 inversion steps, 50 edit steps
FRAMES, LAT = 16, 64          # 16 frames, 512/8 = 64
PEAK_MFMA_F16_TFLOPS = 2500.0  # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)
# context only (frac is always quoted against the nominal peak above): what a stream of nothing but MFMAs sustains on this part with
# random fp16 operands, all 1024 SIMDs busy -- the matrix pipe is power limited (tools/ktile_probe.hip, profiles/r02_ktile_probe.txt)
SUSTAINED_NOTE_32 = {"mfma_only_random_fp16_operands_tflops": 1720, "mfma_shape": "32x32x16", "source": "profiles/r02_ktile_probe.txt"}
SUSTAINED_NOTE_16 = {"mfma_only_random_fp16_operands_tflops": 1920, "mfma_shape": "16x16x32", "source": "profiles/r02_ktile_probe.txt"}


def synthetic_clip(device, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.float16).to(device)
    lat = r(1, 4, FRAMES, LAT, LAT)
    ehs = r(3, 77, 1024)                      # [ddim_inv prompt, negative, edit] (pipeline_i2vgen_xl.py:1044)
    ie = r(3, 1, 1024)
    ie[1].zero_()                             # zero negative image embedding (:437-439)
    il = r(2, 4, FRAMES, LAT, LAT)            # first-frame latents of the source / edited frame
    for i in range(1, FRAMES):                # frame-position planes (:548-554)
        il[:, :, i] = i / (FRAMES - 1)
    il_all = torch.stack([il[0], il[1], il[1]]).contiguous()
    return lat, ehs, ie, il_all


def measure_kernel(fn, iters=20, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms per launch, HIP events on the launching stream


def measured_traffic(kernel_key):
    """HBM bytes per launch of a kernel from the newest profiles/r*_traffic.json -- written by tools/pmc_traffic.py from the
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (kernel name, git commit and the raw counters are recorded there).  The bench
    cannot collect PMC counters itself (they need a rocprofv3 pass per counter group), so the value is `null` unless such a file
    names this kernel; it is never a literal in this file."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    def readable(name):
        """rocprofv3 leaves some kernel names mangled: _Z24flash_attn_d64_v2_kernelILi3ELi1ELi8EEv... -> flash_attn_d64_v2_kernel<3, 1, 8>"""
        import re
        m = re.match(r"_Z(\d+)", name)
        if not m:
            return name
        n = int(m.group(1))
        base, rest = name[m.end():m.end() + n], name[m.end() + n:]
        args = re.findall(r"L([ib])(\d+)E", rest.split("Ev")[0]) if rest.startswith("I") else []
        return base + ("<" + ", ".join(("true" if v == "1" else "false") if t == "b" else v for t, v in args) + ">" if args else "")

    try:
        rec = json.load(open(files[-1]))
        for k, v in rec.get("kernels", {}).items():
            if kernel_key in readable(k):
                return int(v["hbm_bytes_per_launch"]), {"file": os.path.relpath(files[-1], ROOT), "commit": rec.get("commit"),
                                                        "kernel": k}
    except Exception:
        pass
    return None, None


def measured_mfma_busy(kernel_key):
    """Matrix-pipe utilisation of a kernel from the newest profiles/r*_pmc.json -- written by tools/pmc_sq.py from a rocprofv3 SQ counter
    pass (SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE; git commit, formulae and raw counters are in that file).  Like the HBM
    traffic, a stored figure (the bench cannot run a PMC pass inside itself): labelled with its source, `null` when no file names the kernel."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None, None

    def readable(name):
        m = re.match(r"_Z(\d+)", name)
        if not m:
            return name
        n = int(m.group(1))
        base, rest = name[m.end():m.end() + n], name[m.end() + n:]
        args = re.findall(r"L([ib])(\d+)E", rest.split("Ev")[0]) if rest.startswith("I") else []
        return base + ("<" + ", ".join(("true" if v == "1" else "false") if t == "b" else v for t, v in args) + ">" if args else "")
    try:
        rec = json.load(open(files[-1]))
        for k, v in rec.get("kernels", {}).items():
            if kernel_key in readable(k):
                return v["mfma_busy_frac"], {"file": os.path.relpath(files[-1], ROOT), "commit": rec.get("commit"), "kernel": k,
                                             "mfma_busy_frac_vs_sq_busy": v.get("mfma_busy_frac_sq"), "effective_clock_ghz": v.get("effective_clock_ghz")}
    except Exception:
        pass
    return None, None


def _with_pmc(r, kernel_key):
    r["mfma_busy_frac"], r["mfma_busy_source"] = measured_mfma_busy(kernel_key)
    return r


def roofline_spatial_attention(device, pnp=False):
    """Spatial self-attention at the PnP-step shape (N=48 images, 5 heads, S=4096, d=64): 4*N*h*S^2*d FLOP.

    pnp=False: the plain launch (every branch its own Q, K, V) -- flash_attn_d64_v2_kernel<3,1,8> (8-wave blocks); algorithmic
    FLOP == executed FLOP.  `traffic` is the HBM byte count per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) read from
    the newest profiles/r*_traffic.json (see measured_traffic), or null.
    pnp=True: the launch the edit loop actually issues on injection steps (Q/K of all three branches alias the source
    branch) -- flash_attn_d64_v2_kernel<3,3,8> shares one S/softmax over three V streams, so it executes 2/3 of the
    reference op's MFMA FLOP; `achieved` prices the reference op's algorithmic FLOP (as the contract defines it) and
    `executed_tflops` the MFMA work the kernel really issues."""
    from anyv2v_amd import ops
    N, h, S, d = 48, 5, 4096, 64
    C = h * d
    qkv = (torch.randn(N * S, 3 * C, device=device) * 1.0).to(torch.float16)
    o = torch.empty(N * S, C, dtype=torch.float16, device=device)
    fn = lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=N, heads=h, Sq=S, Sk=S, inner=1,
                               q_strides=(S, 0, 1), kv_strides=(S, 0, 1), qk_mod=N // 3 if pnp else 0)
    ms = measure_kernel(fn)
    flops = 4.0 * N * h * S * S * d
    ach = flops / (ms * 1e-3) / 1e12
    traffic, src = measured_traffic("flash_attn_d64_v2_kernel<3, 3, 8>" if pnp else "flash_attn_d64_v2_kernel<3, 1, 8>")
    r = {"bound": "mfma", "kernel": ("flash_attn_d64_v2_kernel<3,3,8> (spatial self-attn under PnP q/k injection, shared softmax, 8-wave blocks, "
                                     if pnp else "flash_attn_d64_v2_kernel<3,1,8> (spatial self-attn, ") + "N=48 h=5 S=4096 d=64)",
         "achieved": round(ach, 2), "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F16_TFLOPS, 4),
         "ms_per_launch": round(ms, 4), "flops_per_launch": flops, "traffic": traffic, "traffic_source": src}
    if pnp:
        r["executed_tflops"] = round(ach * 2.0 / 3.0, 2)
    r["context"] = SUSTAINED_NOTE_32
    return _with_pmc(r, "flash_attn_d64_v2_kernel<3, 3, 8>" if pnp else "flash_attn_d64_v2_kernel<3, 1, 8>")


def roofline_conv(device):
    """ResNet conv3x3 320->320 at 64x64, N=48 (the most frequent conv shape): 2*T*9*Cin*Cout FLOP."""
    from anyv2v_amd import ops
    N, H, C = 48, 64, 320
    x = torch.randn(N * H * H, C, device=device).to(torch.float16)
    w = (torch.randn(C, 9 * C, device=device) / (9 * C) ** 0.5).to(torch.float16)
    b = torch.zeros(C, dtype=torch.float16, device=device)
    out = torch.empty(N * H * H, C, dtype=torch.float16, device=device)
    fn = lambda: ops.gemm(x, w, bias=b, mode=ops.MODE_CONV2D, conv=(H, H, H, H, 1, 0), out=out)
    ms = measure_kernel(fn)
    flops = 2.0 * N * H * H * 9 * C * C
    ach = flops / (ms * 1e-3) / 1e12
    traffic, src = measured_traffic("gemm_big_kernel<3, false, 1")
    return _with_pmc({"bound": "mfma", "kernel": "gemm_big_kernel<3,false,conv2d> (conv3x3 320->320 @64x64, N=48; persistent 192x320 tiles)", "achieved": round(ach, 2),
            "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F16_TFLOPS, 4),
            "ms_per_launch": round(ms, 4), "flops_per_launch": flops, "traffic": traffic, "traffic_source": src,
            "context": SUSTAINED_NOTE_16}, "gemm_big_kernel<3, false, 1")


def roofline_gemm_ws(device):
    """Transformer feed-forward up-projection + GEGLU at the edit step's 64x64 level (3 branches x 16 frames x 4096 = 196608 tokens, 320 -> 2 x 1280): the
    weight-stationary kernel (gemm_ws.hip).  2*M*K*N FLOP with N = 2560 (both halves of the GEGLU projection)."""
    from anyv2v_amd import ops
    M, K, N = 196608, 320, 2560
    x = torch.randn(M, K, device=device).to(torch.float16)
    w = (torch.randn(N, K, device=device) / K ** 0.5).to(torch.float16)
    b = torch.zeros(N, dtype=torch.float16, device=device)
    out = torch.empty(M, N // 2, dtype=torch.float16, device=device)
    fn = lambda: ops.gemm(x, w, bias=b, act=ops.ACT_GEGLU, out=out)
    ms = measure_kernel(fn)
    flops = 2.0 * M * K * N
    ach = flops / (ms * 1e-3) / 1e12
    traffic, src = measured_traffic("gemm_ws_kernel<320, 160, true")
    return _with_pmc({"bound": "mfma", "kernel": "gemm_ws_kernel<K=320, GEGLU> (feed-forward up-projection + GEGLU, 196608 x 320 -> 1280; "
                                       "weight slab resident in LDS, wave-private row strips)", "achieved": round(ach, 2),
            "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F16_TFLOPS, 4),
            "ms_per_launch": round(ms, 4), "flops_per_launch": flops, "traffic": traffic, "traffic_source": src,
            "algorithmic_bytes_per_launch": M * K * 2 + N * K * 2 + M * (N // 2) * 2, "context": SUSTAINED_NOTE_16}, "gemm_ws_kernel<320, 160, true")


def roofline_ff(device):
    """The fused feed-forward launch of the edit step's 64x64 level (ff_fused.hip): GEGLU up-projection 320 -> 2 x 1280, down-projection
    1280 -> 320 and the residual in ONE kernel; 2 M (2 H C + H C) FLOP, algorithmic bytes = x in, residual in, y out + the weights
    (the [M, 1280] hidden activation never reaches HBM)."""
    from anyv2v_amd import ops
    M, C, H = 196608, 320, 1280
    x = torch.randn(M, C, device=device).to(torch.float16)
    r = torch.randn(M, C, device=device).to(torch.float16)
    w1 = (torch.randn(2 * H, C, device=device) / C ** 0.5).to(torch.float16)
    b1 = torch.zeros(2 * H, dtype=torch.float16, device=device)
    w2s = ops.ff_pack_w2((torch.randn(C, H, device=device) / H ** 0.5).to(torch.float16))
    b2 = torch.zeros(C, dtype=torch.float16, device=device)
    out = torch.empty(M, C, dtype=torch.float16, device=device)
    fn = lambda: ops.ff_geglu(x, w1, b1, w2s, b2, residual=r, out=out)
    ms = measure_kernel(fn)
    flops = 2.0 * M * (2 * H * C + H * C)
    ach = flops / (ms * 1e-3) / 1e12
    traffic, src = measured_traffic("ff_fused_c320_kernel<true")
    return _with_pmc({"bound": "mfma", "kernel": "ff_fused_c320_kernel (feed-forward of the 320-channel blocks: GEGLU up-projection + down-projection + "
                                       "residual, 196608 tokens; weights streamed by LDS-DMA, hidden activation in registers)",
            "achieved": round(ach, 2), "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F16_TFLOPS, 4),
            "ms_per_launch": round(ms, 4), "flops_per_launch": flops, "traffic": traffic, "traffic_source": src,
            "algorithmic_bytes_per_launch": 3 * M * C * 2 + 3 * H * C * 2, "context": SUSTAINED_NOTE_16}, "ff_fused_c320_kernel<true")


def effective_cpus() -> int:
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota).  The GPU box shows 256 hardware
    threads but runs under a 16-CPU cgroup quota; 256 torch threads there get CFS-throttled to a crawl."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline():
    """The oracle (fp32 PyTorch restatement of the reference UNet + the PnP hooks) timed on this host's cores on a
    bounded sample: ONE inversion step (B=1) + ONE PnP step (B=3, all hooks on) of BASELINE config 1
    (1 clip x 8 frames x 256x256), extrapolated to the 50+50-step job: frames/s = 8 / (50 * (t1 + t3)).  About 30 s on 16 cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pnp_oracle
    from oracle.unet_oracle import UNetConfig, build_random_oracle
    import gpu_checks as gc
    cores = effective_cpus()
    torch.set_num_threads(cores)
    cfg = UNetConfig.i2vgen_xl()
    oracle = build_random_oracle(cfg, 0)
    inp = gc.config1_inputs(cfg, 3, 8, 32)
    kw = lambda s: dict(fps=inp["fps"][s], image_latents=inp["image_latents"][s], image_embeddings=inp["image_embeddings"][s],
                        encoder_hidden_states=inp["encoder_hidden_states"][s])
    def timed(fn, reps):
        ts = []
        for _ in range(reps):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        return sorted(ts)

    with torch.no_grad():  # one untimed warm-up, then the median of 3, for both step kinds
        f1 = lambda: oracle(inp["sample"][:1], 981, **kw(slice(0, 1)))
        f1()
        t1 = timed(f1, 3)[1]
        pnp_oracle.init_pnp(oracle, 50, 1.0, 1.0, 1.0)
        pnp_oracle.register_time(oracle, 981)
        f3 = lambda: oracle(inp["sample"], 981, **kw(slice(0, 3)))
        f3()
        t3 = timed(f3, 3)[1]
    fps = 8.0 / (STEPS_PER_STAGE * (t1 + t3))
    return {"value": round(fps, 6), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"config 1 (1 clip x 8f x 256x256): 1 inversion step B=1 ({t1:.2f}s) + 1 PnP step B=3 with hooks ({t3:.2f}s), "
                      f"each 1 warm-up + median of 3, fp32 torch on {cores} threads, extrapolated x50 steps per stage",
            "seconds_B1": round(t1, 3), "seconds_B3": round(t3, 3)}


def whole_clip(pipe, device, seed):
    """ONE whole BASELINE config-3 clip through the product pipeline, end to end on the GPU, next to the steady-state number:
    native AutoencoderKL encode of 16 synthetic 512x512 frames (+ first-frame latent) -> pipe.invert (50 steps, trajectory
    written to ddim_latents_{t}.pt files) -> pipe.sample_with_pnp (50 steps, all three injections on every step and
    ddim_init_latents_t_idx 0: the active demo entry, configs/group_pnp_edit/group_config.json:2-13 -- the same schedule as the
    steady-state number; cfg 9.0) -> native VAE decode to 16 PIL frames.  CLIP text / image embeddings
    are synthetic tensors (the towers are not part of this timing).  Run twice: the first clip pays HIP-graph capture."""
    import shutil
    import tempfile
    import numpy as np
    from PIL import Image
    from anyv2v_amd import pnp_utils
    from anyv2v_amd.encoders import attach_native_vae
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    from anyv2v_amd.utils import wait_for_pending_writes
    attach_native_vae(pipe, random_init_seed=0)
    rng = np.random.default_rng(seed)
    frames = [Image.fromarray(rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)) for _ in range(FRAMES)]
    g = torch.Generator().manual_seed(seed)
    r = lambda *sh: torch.randn(*sh, generator=g).to(torch.float16).to(device)
    ehs, ie = r(3, 77, 1024), r(3, 1, 1024)
    out = {}
    for tag in ("first", "second"):
        tmp = tempfile.mkdtemp(prefix="anyv2v_bench_clip_")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lat = pipe.encode_vae_video(frames, device, height=512, width=512)
        first = pipe.vae.encode_image(frames[0], device, 512, 512)
        il = pipe.prepare_image_latents_from_first_frame_latent(first, FRAMES)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pipe.register_modules(scheduler=DDIMInverseScheduler())
        traj = pipe.invert(prompt_embeds=ehs[:1], image_embeddings=ie[:1], image_latents=il, height=512, width=512,
                           num_frames=FRAMES, num_inference_steps=STEPS_PER_STAGE, guidance_scale=1.0, target_fps=8, latents=lat,
                           output_dir=tmp, background_save=True, return_trajectory=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        sched = DDIMScheduler()
        sched.set_timesteps(STEPS_PER_STAGE)
        pipe.register_modules(scheduler=sched)
        k = lambda ratio: sched.timesteps[: int(STEPS_PER_STAGE * ratio)]
        pnp_utils.register_conv_injection(pipe, k(1.0))
        pnp_utils.register_spatial_attention_pnp(pipe, k(1.0))
        pnp_utils.register_temp_attention_pnp(pipe, k(1.0))
        T = max(traj.keys())
        video = pipe.sample_with_pnp(prompt_embeds=ehs[2:3], negative_prompt_embeds=ehs[1:2], image_embeddings=ie[2:3],
                                     image_latents=il, height=512, width=512, num_frames=FRAMES,
                                     num_inference_steps=STEPS_PER_STAGE, guidance_scale=9.0, target_fps=8, latents=traj[T].clone(),
                                     output_type="pil", decode_chunk_size=1, ddim_init_latents_t_idx=0, ddim_inv_latents_path=traj,
                                     ddim_inv_prompt_embeds=ehs[:1], ddim_inv_image_embeddings=ie[:1],
                                     ddim_inv_image_latents=il).frames[0]
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        wait_for_pending_writes(tmp)
        t4 = time.perf_counter()
        pnp_utils.clear_time(pipe)
        nfiles = len([f for f in os.listdir(tmp) if f.startswith("ddim_latents_")])
        shutil.rmtree(tmp, ignore_errors=True)
        out[tag] = {"seconds": round(t4 - t0, 3), "frames_per_s": round(FRAMES / (t4 - t0), 4), "vae_encode_s": round(t1 - t0, 3),
                    "invert_50_steps_s": round(t2 - t1, 3), "pnp_edit_50_steps_plus_vae_decode_s": round(t3 - t2, 3),
                    "trajectory_files_flush_s": round(t4 - t3, 3), "trajectory_files": nfiles, "frames_out": len(video)}
    out["what"] = ("one config-3 clip end to end: native VAE encode (16 x 512x512) + pipe.invert 50 steps (+ddim_latents_t.pt files) + "
                   "pipe.sample_with_pnp 50 steps (injection schedules 1.0/1.0/1.0 as the active demo entry, cfg 9) + native VAE decode; synthetic CLIP embeddings; "
                   "'first' includes HIP-graph capture")
    return out


ALGORITHMIC_TFLOP = {"B1": 20.935, "B3": 62.805}   # torch flop counter over the reference architecture, 16 f x 512^2


def executed_flops(engine, hooks_t=None):
    """MFMA work the product path really issues in ONE forward of ``engine`` (vs the algorithmic count, which prices work the edit
    step skips exactly: V-only projections, shared softmax, shared stem, source-only conv path, hoisted conditioning): every
    ops.gemm / ops.attention launch of one eager forward is recorded -- 2 M N K per GEMM (taps and both GEGLU halves included),
    4 b h Sq Sk d per attention launch, x 2/3 when the spatial shared-softmax kernel runs it (one S / softmax for three V)."""
    from anyv2v_amd import ops, pnp_utils
    acc = {"gemm": 0.0, "attention": 0.0, "launches": 0}
    g0, a0 = ops.gemm, ops.attention

    def gemm(a, w, *args, **kw):
        M = kw.get("M") or a.shape[0]
        acc["gemm"] += 2.0 * M * w.shape[0] * w.shape[1]
        acc["launches"] += 1
        return g0(a, w, *args, **kw)

    def attention(q, k, v, out, **kw):
        fl = 4.0 * kw["batch"] * kw["heads"] * kw["Sq"] * kw["Sk"] * kw.get("head_dim", 64)
        if kw.get("qk_mod", 0) and kw["Sq"] > 16 and not (ops.ATTN_FLAGS & 8):
            fl *= 2.0 / 3.0
        acc["attention"] += fl
        acc["launches"] += 1
        return a0(q, k, v, out, **kw)

    f0 = ops.ff_geglu

    def ff_geglu(x, w1p, b1p, w2s, b2, **kw):   # fused feed-forward: up-projection (both GEGLU halves) + down-projection
        H = w2s.shape[0] * 32
        acc["gemm"] += 2.0 * x.shape[0] * (2 * H * x.shape[1] + H * x.shape[1])
        acc["launches"] += 1
        return f0(x, w1p, b1p, w2s, b2, **kw)

    ops.gemm, ops.attention, ops.ff_geglu = gemm, attention, ff_geglu
    try:
        if hooks_t is not None:
            pnp_utils.register_time(engine.pipe, hooks_t)
        else:
            pnp_utils.clear_time(engine.pipe)
        engine.unet._forward_core(engine.ctx, engine.sample.clone(), drop_source_tail=getattr(engine, "drop_src_tail", False))
        torch.cuda.synchronize()
    finally:
        ops.gemm, ops.attention, ops.ff_geglu = g0, a0, f0
    return acc


def clip_towers_timing(device):
    """The native CLIP towers at the checkpoint's sizes (OpenCLIP ViT-H/14: text 24 x 1024 / 16 heads, 77 tokens; vision 32 x 1280 /
    16 heads x 80, 257 tokens), random weights: what one clip pays for them -- three prompts (editing, negative, inversion prompt;
    pipeline_i2vgen_xl.py:1027-1044) and two 224^2 images (source and edited first frame, :1069-1091).  Their attention is the
    generic one-thread-per-query kernel (anyv2v_attention_small_f16)."""
    from anyv2v_amd.clip import CLIPTextTower, CLIPTowerConfig, CLIPVisionTower

    def rnd_sd(prefix, H, L, I, extra):
        g = torch.Generator(device=device).manual_seed(0)
        r = lambda *sh: (torch.randn(*sh, generator=g, device=device) * 0.02).half()
        sd = dict(extra(r))
        for i in range(L):
            b = f"{prefix}.encoder.layers.{i}."
            for n in ("q", "k", "v", "out"):
                sd[b + f"self_attn.{n}_proj.weight"], sd[b + f"self_attn.{n}_proj.bias"] = r(H, H), r(H)
            sd[b + "mlp.fc1.weight"], sd[b + "mlp.fc1.bias"], sd[b + "mlp.fc2.weight"], sd[b + "mlp.fc2.bias"] = r(I, H), r(I), r(H, I), r(H)
            for n in ("layer_norm1", "layer_norm2"):
                sd[b + n + ".weight"], sd[b + n + ".bias"] = 1 + r(H), r(H)
        return sd

    tcfg = CLIPTowerConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, vocab_size=49408,
                           max_position_embeddings=77)
    vcfg = CLIPTowerConfig(hidden_size=1280, num_hidden_layers=32, num_attention_heads=16, intermediate_size=5120, image_size=224,
                           patch_size=14, projection_dim=1024)
    text = CLIPTextTower(tcfg, rnd_sd("text_model", 1024, 24, 4096, lambda r: {
        "text_model.embeddings.token_embedding.weight": r(49408, 1024), "text_model.embeddings.position_embedding.weight": r(77, 1024),
        "text_model.final_layer_norm.weight": 1 + r(1024), "text_model.final_layer_norm.bias": r(1024)})).to(device)
    vis = CLIPVisionTower(vcfg, rnd_sd("vision_model", 1280, 32, 5120, lambda r: {
        "vision_model.embeddings.patch_embedding.weight": r(1280, 3, 14, 14), "vision_model.embeddings.class_embedding": r(1280),
        "vision_model.embeddings.position_embedding.weight": r(257, 1280), "vision_model.pre_layrnorm.weight": 1 + r(1280),
        "vision_model.pre_layrnorm.bias": r(1280), "vision_model.post_layernorm.weight": 1 + r(1280),
        "vision_model.post_layernorm.bias": r(1280), "visual_projection.weight": r(1024, 1280)})).to(device)
    ids = torch.randint(0, 49408, (3, 77), device=device)
    px = torch.randn(2, 3, 224, 224, device=device).half()
    out = {}
    for name, fn in (("text_3_prompts_ms", lambda: text.encode_ids(ids, 1)), ("vision_2_images_ms", lambda: vis.image_embeds(px))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            y = fn()
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
        assert bool(torch.isfinite(y.float()).all())
    out["total_ms"] = round(out["text_3_prompts_ms"] + out["vision_2_images_ms"], 2)
    out["what"] = ("native CLIP towers (anyv2v_amd/clip.py) at the checkpoint's sizes, random weights, eager launches: 3 prompts x 77 tokens "
                   "through 24 layers (clip_skip 1) + 2 images x 257 tokens through 32 layers; once per clip, not in `seconds` above")
    return out


def multi_edit(pipe, device, seed):
    """Several edits of ONE clip (the demo group config holds 8 edits of one clip): 50-step inversion once, then 50-step PnP edits
    with different prompts / edited frames, all injections on -- without and with ``pipeline.SourceFeatureCache`` (the first edit
    records the source branch's injected features, every further edit replays them and runs [negative, editing] only; outputs
    bit-equal, tests).  Separate from the headline metric, which stays a single edit."""
    from anyv2v_amd import pnp_utils
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    g = torch.Generator().manual_seed(seed)
    r = lambda *sh: torch.randn(*sh, generator=g).to(torch.float16).to(device)
    lat, ehs, ie, il_all = synthetic_clip(device, seed)
    pipe.register_modules(scheduler=DDIMInverseScheduler())
    traj = pipe.invert(prompt_embeds=ehs[:1], image_embeddings=ie[:1], image_latents=il_all[:1], height=512, width=512,
                       num_frames=FRAMES, num_inference_steps=STEPS_PER_STAGE, guidance_scale=1.0, target_fps=8, latents=lat,
                       return_trajectory=True)
    T = max(traj.keys())
    edits = [(r(1, 77, 1024), r(1, 1, 1024)) for _ in range(3)]

    def edit(k):
        sched = DDIMScheduler()
        sched.set_timesteps(STEPS_PER_STAGE)
        pipe.register_modules(scheduler=sched)
        pnp_utils.register_conv_injection(pipe, sched.timesteps)
        pnp_utils.register_spatial_attention_pnp(pipe, sched.timesteps)
        pnp_utils.register_temp_attention_pnp(pipe, sched.timesteps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pipe.sample_with_pnp(prompt_embeds=edits[k][0], negative_prompt_embeds=ehs[1:2], image_embeddings=edits[k][1],
                                   image_latents=il_all[1:2], height=512, width=512, num_frames=FRAMES,
                                   num_inference_steps=STEPS_PER_STAGE, guidance_scale=9.0, target_fps=8, latents=traj[T].clone(),
                                   output_type="latent", ddim_init_latents_t_idx=0, ddim_inv_latents_path=traj,
                                   ddim_inv_prompt_embeds=ehs[:1], ddim_inv_image_embeddings=ie[:1],
                                   ddim_inv_image_latents=il_all[:1]).frames
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pnp_utils.clear_time(pipe)
        return dt, out

    pipe.enable_source_cache(False)
    edit(0)                                   # graph capture
    plain = [edit(k) for k in range(3)]
    cache = pipe.enable_source_cache(True)
    cached = [edit(k) for k in range(3)]      # edit 0 records (+ captures the record / replay graphs on first use), 1 and 2 replay
    cached2 = [edit(k) for k in range(3)]     # steady state: everything replayed
    equal = all(bool(torch.equal(a[1], b[1])) for a, b in zip(plain, cached)) and all(bool(torch.equal(a[1], b[1])) for a, b in zip(plain, cached2))
    md = lambda a, b: float((a[1].float() - b[1].float()).abs().max())
    res = {"edit_seconds_without_cache": [round(x[0], 3) for x in plain],
           "edit_seconds_with_cache_first_pass": [round(x[0], 3) for x in cached],
           "edit_seconds_with_cache_replay": [round(x[0], 3) for x in cached2],
           "speedup_per_additional_edit": round(sum(x[0] for x in plain) / sum(x[0] for x in cached2), 3),
           "bit_equal_to_uncached": equal, "max_abs_diff_first_pass": [md(a, b) for a, b in zip(plain, cached)],
           "max_abs_diff_replay_pass": [md(a, b) for a, b in zip(plain, cached2)], "cache_gib": round(cache.nbytes() / 2**30, 2),
           "recorded_steps": cache.recorded_steps, "replayed_steps": cache.replayed_steps,
           "what": "50-step PnP edits (cfg 9, conv + spatial + temporal injection on every step) of one inverted 16 f x 512^2 clip, "
                   "latents out; first pass: edit 0 records the source branch's features (three-branch steps + copies), edits 1-2 replay "
                   "(two-branch steps); replay pass: all three replayed"}
    pipe.enable_source_cache(False)
    return res


def config5_steps(pipe, device, seed, frames=128):
    """BASELINE config 5 (long-video mode: 1 clip x 128 frames x 512x512, one GPU): one inversion step (UNet B=1) and one PnP edit step
    (UNet B=3, conv + spatial + temporal injection on) of the full UNet at full size, HIP-graph replays, median of 3 after the capture
    step.  Temporal attention runs over all 128 frames (the strided flash kernel); the parity of this size is tests/gpu_checks.py
    check_config5_full_size."""
    from anyv2v_amd import pnp_utils
    from anyv2v_amd.pipeline import _StepEngine
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    g = torch.Generator().manual_seed(seed)
    r = lambda *sh: torch.randn(*sh, generator=g).to(torch.float16).to(device)
    lat, ehs, ie, il = r(1, 4, frames, LAT, LAT), r(3, 77, 1024), r(3, 1, 1024), r(2, 4, frames, LAT, LAT)
    ie[1].zero_()
    for i in range(1, frames):
        il[:, :, i] = i / (frames - 1)
    il_all = torch.stack([il[0], il[1], il[1]]).contiguous()
    inv, fwd = DDIMInverseScheduler(), DDIMScheduler()
    inv.set_timesteps(STEPS_PER_STAGE)
    fwd.set_timesteps(STEPS_PER_STAGE)
    ts_inv, ts_pnp = [int(t) for t in inv.timesteps], [int(t) for t in fwd.timesteps]
    pnp_utils.register_conv_injection(pipe, fwd.timesteps)
    pnp_utils.register_spatial_attention_pnp(pipe, fwd.timesteps)
    pnp_utils.register_temp_attention_pnp(pipe, fwd.timesteps)
    s_inv, s_pnp = lat.clone(), lat.repeat(3, 1, 1, 1, 1).contiguous()
    cond1 = dict(encoder_hidden_states=ehs[:1].contiguous(), fps=torch.tensor([8], device=device),
                 image_latents=il_all[:1].contiguous(), image_embeddings=ie[:1].contiguous())
    cond3 = dict(encoder_hidden_states=ehs, fps=torch.tensor([8, 8, 8], device=device), image_latents=il_all, image_embeddings=ie)
    out = {}
    torch.cuda.reset_peak_memory_stats()   # (the figure below is this function's peak: weights + the 128-frame activations and graphs)
    for name, mk, tt, cf, key, reg, smp in (
            ("inversion_step_B1_ms", lambda: _StepEngine(pipe, s_inv, cond1, b_unc=-1, b_cond=0, guidance=1.0, dup_slots=[]),
             torch.tensor(ts_inv, dtype=torch.float32, device=device)[:, None].contiguous(), inv.coefficient_table(ts_inv, device),
             ("inv",), None, s_inv),
            ("pnp_edit_step_B3_ms", lambda: _StepEngine(pipe, s_pnp, cond3, b_unc=1, b_cond=2, guidance=9.0, dup_slots=[1],
                                                        shared_stem=True),
             torch.tensor(ts_pnp, dtype=torch.float32, device=device)[:, None].expand(-1, 3).contiguous(),
             fwd.coefficient_table(ts_pnp, device), ("pnp",), ts_pnp, s_pnp)):
        if reg is None:
            pnp_utils.clear_time(pipe)
        else:
            pnp_utils.register_time(pipe, reg[0])
        eng = mk()
        if reg is not None:
            eng.drop_src_tail = True
        eng.step(tt[0], cf[0], key=key + (pnp_utils.injection_state(pipe) if reg is not None else ()))   # graph capture
        torch.cuda.synchronize()
        times = []
        for j in (1, 2, 3):
            t0 = time.perf_counter()
            eng.step(tt[j], cf[j], key=key + (pnp_utils.injection_state(pipe) if reg is not None else ()))
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        out[name] = round(sorted(times)[1], 2)
        out[name.replace("_ms", "_finite")] = bool(torch.isfinite(smp.float()).all())
        del eng
        torch.cuda.empty_cache()
    pnp_utils.clear_time(pipe)
    pair_ms = out["inversion_step_B1_ms"] + out["pnp_edit_step_B3_ms"]
    out["frames_per_s"] = round(frames / STEPS_PER_STAGE / (pair_ms * 1e-3), 4)
    out["peak_hbm_gib"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
    out["what"] = (f"BASELINE config 5: 1 clip x {frames} f x 512x512 on one GPU; one inversion step + one PnP edit step at full size (graph "
                   f"replays, median of 3); frames_per_s = {frames} / (50 x their sum) -- the 50-step loops are not run in the bench")
    return out


def finish_distributed(dist, dt, latents, world, device):
    """The one collective of the sharded job -- all_gather of every rank's edited latents (512 KiB per rank at
    16f x 512^2; RCCL over xGMI on the GPU node, gloo in the CPU test) -- plus MAX over ranks of the timed region."""
    out = [torch.empty_like(latents) for _ in range(world)]
    dist.all_gather(out, latents)
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return float(tmax.item()), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-clip", action="store_true", help="skip the end-to-end timing of one whole clip")
    ap.add_argument("--no-multi-edit", action="store_true", help="skip the several-edits-of-one-clip timing (source feature cache)")
    ap.add_argument("--seed", type=int, default=8888)
    ap.add_argument("--pipelined", action="store_true",
                    help="time the job-level two-clip software pipeline (inversion step of the NEXT clip beside the edit step of the "
                         "current clip, two streams, as run_group_anyv2v runs a multi-clip job) instead of the single-clip serial order; "
                         "by default the pipelined rate is measured after the timed region and reported as config.pipelined_ms_per_step")
    ap.add_argument("--serial", action="store_true", help="(default since round 5; kept for old command lines)")
    ap.add_argument("--overlap", dest="pipelined", action="store_true", help=argparse.SUPPRESS)   # rounds 3-4 name of --pipelined
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (BASELINE configs 2 and 5)")
    ap.add_argument("--no-job-schedule", action="store_true",
                    help="skip the measurement of the other schedule behind the timed region (profiling runs: the trace then holds the timed pairs only)")
    args = ap.parse_args()
    args.overlap = args.pipelined and not args.serial   # (the timed schedule; also the top-level "schedule" key of the line)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    from anyv2v_amd import pnp_utils
    from anyv2v_amd.pipeline import I2VGenXLPipeline, _StepEngine
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler

    torch.set_grad_enabled(False)
    pipe = I2VGenXLPipeline.from_pretrained("ali-vilab/i2vgen-xl", torch_dtype=torch.float16, variant="fp16",
                                            random_init_seed=0)
    pipe.to(device)
    lat, ehs, ie, il_all = synthetic_clip(device, args.seed + rank)  # one independent clip per rank (weak scaling)
    fps1, fps3 = torch.tensor([8], device=device), torch.tensor([8, 8, 8], device=device)

    inv, fwd = DDIMInverseScheduler(), DDIMScheduler()
    inv.set_timesteps(STEPS_PER_STAGE)
    fwd.set_timesteps(STEPS_PER_STAGE)
    ts_inv, ts_pnp = [int(t) for t in inv.timesteps], [int(t) for t in fwd.timesteps]
    # first demo entry (configs/group_pnp_edit/group_config.json:9-12): all three injections on for every step
    pnp_utils.register_conv_injection(pipe, fwd.timesteps[: int(STEPS_PER_STAGE * 1.0)])
    pnp_utils.register_spatial_attention_pnp(pipe, fwd.timesteps[: int(STEPS_PER_STAGE * 1.0)])
    pnp_utils.register_temp_attention_pnp(pipe, fwd.timesteps[: int(STEPS_PER_STAGE * 1.0)])

    s_inv = lat.clone()
    s_pnp = lat.repeat(3, 1, 1, 1, 1).contiguous()
    cond1 = dict(encoder_hidden_states=ehs[:1].contiguous(), fps=fps1, image_latents=il_all[:1].contiguous(),
                 image_embeddings=ie[:1].contiguous())
    cond3 = dict(encoder_hidden_states=ehs, fps=fps3, image_latents=il_all, image_embeddings=ie)
    pnp_utils.clear_time(pipe)   # inversion steps run hook-free (stage 1 of the reference has no hooks registered)
    e_inv = _StepEngine(pipe, s_inv, cond1, b_unc=-1, b_cond=0, guidance=1.0, dup_slots=[])
    # the edit loop runs on a sibling pipeline object (same weights; its own split-K scratch), as run_group_anyv2v's stage 2 does:
    # the two loops are enqueued on two streams
    e_pnp = _StepEngine(pipe.sibling(ws_slot=1), s_pnp, cond3, b_unc=1, b_cond=2, guidance=9.0, dup_slots=[1], shared_stem=True)
    # as pipe.sample_with_pnp configures its three-branch engine: the source branch's prediction is never read, so its forward stops
    # behind the last hook site (exact, bit-equal; anyv2v_amd/pipeline.py)
    e_pnp.drop_src_tail = os.environ.get("ANYV2V_DROP_SRC_TAIL", "1") == "1"
    tt_inv = torch.tensor(ts_inv, dtype=torch.float32, device=device)[:, None].contiguous()
    tt_pnp = torch.tensor(ts_pnp, dtype=torch.float32, device=device)[:, None].expand(-1, 3).contiguous()
    cf_inv, cf_pnp = inv.coefficient_table(ts_inv, device), fwd.coefficient_table(ts_pnp, device)
    traj = torch.zeros(STEPS_PER_STAGE, 4, FRAMES, LAT, LAT, dtype=torch.float16, device=device)  # HBM-resident trajectory

    def pair(i):
        j = i % STEPS_PER_STAGE
        # --- inversion step (pipeline_i2vgen_xl.py:1385-1433); inversion never injects (cfg 1.0, B=1)
        pnp_utils.clear_time(pipe)
        e_inv.step(tt_inv[j], cf_inv[j], key=("inv",))
        traj[j].copy_(s_inv[0])
        # --- PnP edit step (:1131-1179): source latent from the trajectory, 3-way batch, injections on
        s_pnp[0].copy_(traj[STEPS_PER_STAGE - 1 - j] if i >= STEPS_PER_STAGE else traj[j])
        pnp_utils.register_time(pipe, ts_pnp[j])
        e_pnp.step(tt_pnp[j], cf_pnp[j], key=("pnp",) + pnp_utils.injection_state(pipe))

    pair_serial = pair

    def make_pipelined():
        # software pipeline over the clips of a job: clip k + 1 is inverted while clip k is edited.  The edit reads the trajectory
        # of ITS clip (traj_prev, produced by an earlier inversion -- filled here, outside the timed region), the inversion writes
        # the next clip's; the two steps have no data in common and run on two streams.
        for i in range(STEPS_PER_STAGE):
            pnp_utils.clear_time(pipe)
            e_inv.step(tt_inv[i], cf_inv[i], key=("inv",))
            traj[i].copy_(s_inv[0])
        traj_prev = traj.clone()
        s_inv.copy_(lat)
        st_inv, st_pnp = torch.cuda.Stream(), torch.cuda.Stream()
        serialise = [False]
        torch.cuda.synchronize()

        pacing = os.environ.get("ANYV2V_PIPELINE_PACING", "1") == "1"

        def pair_overlapped(i):
            j = i % STEPS_PER_STAGE
            # the edit step is enqueued first and paces the inversion step (run_group_anyv2v.main_pipelined): inversion step j does not
            # start before edit step j does, so the cheaper loop cannot race ahead and leave the edit alone on the chip
            with torch.cuda.stream(st_pnp):
                if serialise[0]:
                    st_pnp.wait_stream(st_inv)
                started = st_pnp.record_event() if pacing else None
                s_pnp[0].copy_(traj_prev[j])
                pnp_utils.register_time(pipe, ts_pnp[j])
                e_pnp.step(tt_pnp[j], cf_pnp[j], key=("pnp",) + pnp_utils.injection_state(pipe))
            with torch.cuda.stream(st_inv):
                if serialise[0]:
                    st_inv.wait_stream(st_pnp)
                elif started is not None:
                    st_inv.wait_event(started)
                pnp_utils.clear_time(pipe)
                e_inv.step(tt_inv[j], cf_inv[j], key=("inv",))
                traj[j].copy_(s_inv[0])
        def check_overlap(n=3):
            """The same n pairs once with the two streams side by side, once serialised: bit-equal latents."""
            outs = []
            for serial in (False, True):
                s_inv.copy_(lat)
                s_pnp.copy_(lat.repeat(3, 1, 1, 1, 1))
                torch.cuda.synchronize()
                serialise[0] = serial
                for i in range(n):
                    pair_overlapped(i)
                serialise[0] = False
                torch.cuda.synchronize()
                outs.append((s_inv.clone(), s_pnp.clone()))
            return bool(torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]))
        return pair_overlapped, check_overlap

    check_overlap = None
    if args.overlap:
        pair, check_overlap = make_pipelined()
    for i in range(args.warmup):
        pair(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        pair(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        dt, _gathered = finish_distributed(dist, dt, s_pnp[2:3].contiguous(), world, device)
    finite = bool(torch.isfinite(s_pnp.float()).all() and torch.isfinite(s_inv.float()).all())
    rccl_ranks = len(_gathered) if dist is not None else None
    serial_ms = pipelined_ms = None
    bit_equal = None
    if world == 1 and not args.no_job_schedule:
        n = min(args.steps, 20)
        if args.overlap:
            # the same pairs one after the other on one stream (the single-clip order), for comparison
            other = pair_serial
        else:
            # the job-level schedule (two clips on two streams), measured AFTER the timed region: it is not BASELINE config 3 (one clip
            # cannot overlap its own inversion with its own edit) and is reported beside the headline, never as `value`
            other, check_overlap = make_pipelined()
        other(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            other(i)
        torch.cuda.synchronize()
        other_ms = (time.perf_counter() - t0) / n * 1e3
        serial_ms, pipelined_ms = (other_ms, dt / args.steps * 1e3) if args.overlap else (dt / args.steps * 1e3, other_ms)
        bit_equal = check_overlap()

    config2 = None
    if world == 1 and not args.no_configs:
        # BASELINE config 2: the 50-step DDIM inversion of one 16 f x 512^2 clip alone (UNet B=1, no hooks), timed as n steps of that loop
        n = min(args.steps, 20)
        s_inv.copy_(lat)
        pnp_utils.clear_time(pipe)
        e_inv.step(tt_inv[0], cf_inv[0], key=("inv",))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            e_inv.step(tt_inv[i], cf_inv[i], key=("inv",))
            traj[i].copy_(s_inv[0])
        torch.cuda.synchronize()
        inv_ms = (time.perf_counter() - t0) / n * 1e3
        config2 = {"inversion_step_B1_ms": round(inv_ms, 3), "steps_timed": n,
                   "frames_per_s": round(FRAMES / STEPS_PER_STAGE / (inv_ms * 1e-3), 4),
                   "what": "BASELINE config 2: DDIM inversion, 50 steps, 1 clip x 16 f x 512x512, fp16, one GPU; frames_per_s = 16 / (50 x "
                           "the measured step of that loop, trajectory write included)"}

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = FRAMES * (args.steps / STEPS_PER_STAGE) * world / dt
        line = {
            "metric": "frames/sec for 16fx512x512 DDIM-inversion+PnP-edit; spatial-attn MFMA % of peak",
            "value": round(value, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic", "schedule": "pipelined" if args.overlap else "serial",
            "config": {"workload": "BASELINE config 3 per GPU: 1 clip x 16f x 512x512, 50-step DDIM inversion (UNet B=1) + 50-step "
                                   "PnP edit (UNet B=3, cfg 9.0, conv+spatial+temporal injection on every step); a bench step = "
                                   "1 inversion step + 1 edit step = 1/50 clip; I2VGen-XL 3D-UNet 1.42B params, random init",
                       "steps_per_stage": STEPS_PER_STAGE, "frames": FRAMES, "latent": [4, FRAMES, LAT, LAT],
                       "hip_graphs": os.environ.get("ANYV2V_NO_GRAPH", "0") != "1", "finite": finite,
                       "schedule": ("pipelined (--pipelined): the job's steady state -- the inversion step of clip k + 1 and the edit step of "
                                    "clip k on two HIP streams; NOT BASELINE config 3 (which is one clip)") if args.overlap
                       else ("serial: one clip -- inversion step, then edit step, one stream (SURVEY 8(d): frames/s = 16 / seconds for "
                             "one clip; value = 16 / 50 / ms_per_step)"),
                       # the job-level extra: a multi-clip job on one GPU runs the next clip's inversion beside the current clip's edit
                       # (python -m anyv2v_amd.run_group_anyv2v); bit-equal to the serial order; no BASELINE config feeds two clips to a GPU
                       "serial_ms_per_step": None if serial_ms is None else round(serial_ms, 3),
                       "single_clip_frames_per_s": None if serial_ms is None else round(FRAMES / STEPS_PER_STAGE / (serial_ms * 1e-3), 4),
                       "pipelined_ms_per_step": None if pipelined_ms is None else round(pipelined_ms, 3),
                       "job_frames_per_s": None if pipelined_ms is None else round(FRAMES / STEPS_PER_STAGE / (pipelined_ms * 1e-3), 4),
                       "pipelined_bit_equal_to_serial": bit_equal,
                       "excluded": "`value` is the steady-state loop rate: VAE encode/decode, CLIP encoders and file I/O are outside "
                                   "(SURVEY 8(f) F1/F2); the `clip` object times one whole clip including VAE and the trajectory files"},
        }
        if rccl_ranks is not None:   # how many ranks' edited latents the one all_gather of the job delivered to rank 0
            line["rccl_ranks"] = rccl_ranks
            line["rccl_gathered_finite"] = bool(all(torch.isfinite(g.float()).all() for g in _gathered))
        # what is executed vs what the metric prices: one eager forward of each step kind with every GEMM / attention launch recorded
        ex1 = executed_flops(e_inv)
        ex3 = executed_flops(e_pnp, hooks_t=ts_pnp[0])
        ex_t = (ex1["gemm"] + ex1["attention"] + ex3["gemm"] + ex3["attention"]) / 1e12
        alg_t = ALGORITHMIC_TFLOP["B1"] + ALGORITHMIC_TFLOP["B3"]
        line["flops"] = {"algorithmic_tflop_per_step": alg_t, "executed_tflop_per_step": round(ex_t, 3),
                         "executed_tflop_B1": round((ex1["gemm"] + ex1["attention"]) / 1e12, 3),
                         "executed_tflop_B3": round((ex3["gemm"] + ex3["attention"]) / 1e12, 3),
                         "executed_attention_tflop_B3": round(ex3["attention"] / 1e12, 3),
                         "gemm_attention_launches_B1_B3": [ex1["launches"], ex3["launches"]],
                         "step_frac_algorithmic": round(alg_t / (ms * 1e-3) / PEAK_MFMA_F16_TFLOPS, 4),
                         "step_frac_executed": round(ex_t / (ms * 1e-3) / PEAK_MFMA_F16_TFLOPS, 4),
                         "what": "MFMA work per bench step (1 inversion step B=1 + 1 edit step B=3) / ms_per_step / 2500 TFLOP/s; algorithmic = the "
                                 "reference architecture's count (SURVEY 8(d)), executed = summed over the launches of one eager forward of each kind "
                                 "(exact savings: hoisted conditioning, V-only projections, shared softmax, shared stem, source-only conv path, source branch stopped behind the last hook site)"}
        traffic = sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_step_traffic.json")))
        if traffic:
            try:
                line["step_traffic"] = dict(json.load(open(traffic[-1])), source=os.path.relpath(traffic[-1], ROOT))
            except Exception:
                pass
        if not args.no_roofline:  # rank 0, after the timed region (the other ranks wait at destroy_process_group)
            line["roofline"] = roofline_spatial_attention(device)
            line["roofline_pnp"] = roofline_spatial_attention(device, pnp=True)
            line["roofline_gemm"] = roofline_conv(device)
            line["roofline_gemm_ws"] = roofline_gemm_ws(device)
            line["roofline_ff"] = roofline_ff(device)
        if world == 1 and not args.no_clip:
            del e_inv, e_pnp
            torch.cuda.empty_cache()
            pnp_utils.clear_time(pipe)
            line["clip"] = whole_clip(pipe, device, args.seed)
            line["clip"]["clip_towers"] = clip_towers_timing(device)
            e_inv = e_pnp = None
        if world == 1 and not args.no_multi_edit:
            e_inv = e_pnp = None
            torch.cuda.empty_cache()
            pnp_utils.clear_time(pipe)
            line["multi_edit"] = multi_edit(pipe, device, args.seed)
        if world == 1 and not args.no_configs:
            e_inv = e_pnp = None
            torch.cuda.empty_cache()
            line["configs"] = {"config_2_inversion_only": config2, "config_5_long_video": config5_steps(pipe, device, args.seed),
                               "config_3": "the headline (`value`, `ms_per_step`)",
                               "config_1": "the CPU-runnable case: `cpu_baseline` times it; tests/ hold its parity",
                               "config_4": "8 clips on 8 GPUs = this bench under torchrun --nproc-per-node 8 (one clip per rank + one all_gather)"}
        if world == 1 and not args.no_cpu_baseline:
            del pipe, e_inv, e_pnp
            torch.cuda.empty_cache()
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
