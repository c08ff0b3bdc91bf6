"""Generate tests/golden/*.pt by running the REFERENCE's own code (only works where /root/reference exists).

    python tests/golden/make_golden.py

1. ``pnp_hooks_mini.pt``: the reference's ``i2vgen-xl/pnp_utils.py`` (imported verbatim via
   ``oracle.ref_stubs``) registered on the mini oracle UNet; v-predictions for a 3-way batch at
   timesteps that are on all / some / none of the injection schedules, plus the un-hooked output.
   ``pnp_hooks_full_config1.pt`` (``--full``, ~1 min of CPU): the same at full width -- the 1.42 B-parameter oracle at
   BASELINE config 1 (3 x 8 f x 256^2, weights seed 1234, inputs seed 8888 = ``tests/gpu_checks.config1_inputs``), the
   reference's hooks registered on all 17 sites; v-predictions at t=981 (every site injecting) and t=301 (temporal only).
2. ``inverse_scheduler.pt``: the reference's vendored ``consisti2v/ddim_inverse_scheduler.py``
   constructed with the config logged at ``i2vgen-xl/demo.ipynb:1208-1226``: alphas_cumprod table,
   timesteps for n=50/500, and inverse steps on seeded tensors.
3. ``ref_pipeline_mini.pt`` (``--pipeline``) / ``ref_pipeline_full_config1.pt`` (``--pipeline-full``, ~3 min of CPU): the
   reference's OWN PIPELINE CLASS (``i2vgen-xl/pipelines/pipeline_i2vgen_xl.py`` verbatim via ``oracle.ref_pipeline``) runs
   inversion -> CFG reconstruction -> PnP edit of one synthetic clip around the oracle UNet (fp32, CPU), the reference's
   vendored inverse scheduler and toy VAE / CLIP components: the conditioning tensors its glue code built, the trajectory it
   wrote, the reconstructed and the edited latents.  The -m gpu suite runs the HIP pipeline on the same inputs against it.
4. ``consisti2v_decoder_hooks.pt`` (``--consisti2v``) / ``seine_decoder_hooks.pt`` (``--seine``): the sibling hook families -- see
   ``gen_consisti2v`` / ``gen_seine``.
"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_stubs  # noqa: E402
from oracle.unet_oracle import UNetConfig, build_oracle, random_state_dict  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

MINI_SEED = 1234
INPUT_SEED = 8888  # the reference's seed (configs/group_pnp_edit/template.yaml:4)
B, FR, HW = 3, 4, 8
N_STEPS = 50
PNP = dict(pnp_f_t=0.2, pnp_spatial_attn_t=0.5, pnp_temp_attn_t=0.8)  # -> 10 / 25 / 40 steps


def mini_inputs(cfg):
    g = torch.Generator().manual_seed(INPUT_SEED)
    sample = torch.randn(B, 4, FR, HW, HW, generator=g)
    il = torch.randn(B, 4, FR, HW, HW, generator=g)
    for i in range(1, FR):  # frame-position planes, pipeline_i2vgen_xl.py:548-554
        il[:, :, i] = i / (FR - 1)
    ehs = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    ie = torch.randn(B, 1, cfg.cross_attention_dim, generator=g)
    return dict(sample=sample, image_latents=il, encoder_hidden_states=ehs, image_embeddings=ie,
                fps=torch.tensor([8] * B))


def gen_hooks():
    ref = ref_stubs.load_reference_pnp_utils()
    cfg = UNetConfig.mini()
    unet = build_oracle(cfg, random_state_dict(cfg, MINI_SEED), dtype=torch.float32)
    inp = mini_inputs(cfg)
    kw = dict(fps=inp["fps"], image_latents=inp["image_latents"], image_embeddings=inp["image_embeddings"],
              encoder_hidden_states=inp["encoder_hidden_states"])
    out = {"inputs": inp, "mini_seed": MINI_SEED, "pnp": PNP, "n_steps": N_STEPS}
    with torch.no_grad():
        out["v_nohook_t981"] = unet(inp["sample"], 981, **kw)[0].clone()
        pipe = types.SimpleNamespace(unet=unet)
        ts = torch.arange(N_STEPS).flip(0) * (1000 // N_STEPS) + 1  # 981..1, demo.ipynb:1201-1204
        ref.register_conv_injection(pipe, ts[: int(N_STEPS * PNP["pnp_f_t"])])
        ref.register_spatial_attention_pnp(pipe, ts[: int(N_STEPS * PNP["pnp_spatial_attn_t"])])
        ref.register_temp_attention_pnp(pipe, ts[: int(N_STEPS * PNP["pnp_temp_attn_t"])])
        for t in (981, 701, 301, 101):  # on all three / spatial+temp / temp only / none
            ref.register_time(pipe, t)
            out[f"v_hook_t{t}"] = unet(inp["sample"], t, **kw)[0].clone()
    torch.save(out, os.path.join(HERE, "pnp_hooks_mini.pt"))
    for k, v in out.items():
        if torch.is_tensor(v):
            print(k, tuple(v.shape), float(v.abs().max()))


def gen_hooks_full():
    """Full-width fixture.  Only outputs are stored (2 x 393 KB fp32): weights and inputs are
    re-derived from their seeds by the tests."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_checks as gc
    ref = ref_stubs.load_reference_pnp_utils()
    cfg = UNetConfig.i2vgen_xl()
    unet = build_oracle(cfg, random_state_dict(cfg, MINI_SEED), dtype=torch.float32)
    inp = gc.config1_inputs(cfg, 3, 8, 32, seed=INPUT_SEED)
    inp = {k: (v.half().float() if v.is_floating_point() else v) for k, v in inp.items()}   # the tests feed fp16-rounded inputs
    kw = dict(fps=inp["fps"], image_latents=inp["image_latents"], image_embeddings=inp["image_embeddings"],
              encoder_hidden_states=inp["encoder_hidden_states"])
    out = {"weights_seed": MINI_SEED, "input_seed": INPUT_SEED, "pnp": PNP, "n_steps": N_STEPS, "shape": tuple(inp["sample"].shape)}
    with torch.no_grad():
        pipe = types.SimpleNamespace(unet=unet)
        ts = torch.arange(N_STEPS).flip(0) * (1000 // N_STEPS) + 1
        ref.register_conv_injection(pipe, ts[: int(N_STEPS * PNP["pnp_f_t"])])
        ref.register_spatial_attention_pnp(pipe, ts[: int(N_STEPS * PNP["pnp_spatial_attn_t"])])
        ref.register_temp_attention_pnp(pipe, ts[: int(N_STEPS * PNP["pnp_temp_attn_t"])])
        for t in (981, 301):
            ref.register_time(pipe, t)
            out[f"v_hook_t{t}"] = unet(inp["sample"], t, **kw)[0].clone()
            print(f"v_hook_t{t}", tuple(out[f"v_hook_t{t}"].shape), float(out[f"v_hook_t{t}"].abs().max()))
    torch.save(out, os.path.join(HERE, "pnp_hooks_full_config1.pt"))


def gen_scheduler():
    mod = ref_stubs.load_reference_inverse_scheduler()
    cfgd = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="squaredcos_cap_v2",
                clip_sample=False, set_alpha_to_one=True, steps_offset=1, prediction_type="v_prediction",
                timestep_spacing="leading", rescale_betas_zero_snr=True)  # demo.ipynb:1208-1226
    s = mod.DDIMInverseScheduler(**cfgd)
    out = {"alphas_cumprod": s.alphas_cumprod.clone(), "config": cfgd}
    g = torch.Generator().manual_seed(INPUT_SEED)
    x = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    v = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    out["x"], out["v"] = x, v
    for n in (50, 500):
        s.set_timesteps(n)
        out[f"timesteps_{n}"] = s.timesteps.clone()
        for t in (1, int(s.timesteps[1]), int(s.timesteps[n // 2]), int(s.timesteps[-1])):
            out[f"inv_step_n{n}_t{t}"] = s.step(v, t, x).prev_sample.clone()
    torch.save(out, os.path.join(HERE, "inverse_scheduler.pt"))
    print("alphas_cumprod[0,500,998,999] =", [float(out["alphas_cumprod"][i]) for i in (0, 500, 998, 999)])
    print("timesteps_50[:5] =", out["timesteps_50"][:5].tolist(), " timesteps_500[-3:] =", out["timesteps_500"][-3:].tolist())


# one synthetic clip per job: mini UNet (4 f x 64^2, 4 steps, schedules 0.25 / 0.5 / 0.75) and the full-width UNet at BASELINE
# config 1's size (8 f x 256^2; 3 steps, schedules 0.34 / 0.67 / 1.0 -> 1 / 2 / 3 steps)
REF_PIPELINE_JOBS = {
    "mini": dict(cfg="mini", seed=4321, frames=4, size=64, n_steps=4, ratios=(0.25, 0.5, 0.75), file="ref_pipeline_mini.pt"),
    "full": dict(cfg="full", seed=1234, frames=8, size=256, n_steps=3, ratios=(0.34, 0.67, 1.0), file="ref_pipeline_full_config1.pt"),
}


def ref_pipeline_frames(spec):
    """Smooth synthetic frames (a drifting gradient + texture) and an 'edited' first frame, as PIL images."""
    import numpy as np
    from PIL import Image
    n, size = spec["frames"], spec["size"]
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
    rng = np.random.RandomState(spec["seed"])
    tex = rng.rand(size, size, 3).astype(np.float32)
    frames = []
    for i in range(n):
        ph = i / max(n - 1, 1)
        img = np.stack([0.5 + 0.5 * np.sin(6.3 * (xx + ph)), yy, 0.5 + 0.5 * np.cos(6.3 * (xx * yy + ph))], -1)
        frames.append(Image.fromarray((255 * (0.8 * img + 0.2 * tex)).clip(0, 255).astype("uint8")))
    ed = np.stack([yy, 0.5 + 0.5 * np.sin(9.0 * xx), xx], -1)
    edited = Image.fromarray((255 * (0.8 * ed + 0.2 * tex[::-1])).clip(0, 255).astype("uint8"))
    return frames, edited


def pack_ref_pipeline_job(spec, job):
    """What the GPU box needs of a reference-pipeline job (fp16 tensors; the trajectory stacked in inversion order)."""
    h = lambda x: x.detach().to(torch.float16).contiguous()
    return dict(spec={k: v for k, v in spec.items()}, inv_ts=list(job["inv_ts"]), T=job["T"], lat0=h(job["lat0"]),
                trajectory=h(torch.stack([job["files"][t] for t in job["inv_ts"]])), src_pe=h(job["src_pe"]), src_ie=h(job["src_ie"]),
                src_il=h(job["src_il"]), pe=h(job["pe"]), npe=h(job["npe"]), ie_pos=h(job["ie2"][1:]), il_edit=h(job["il2"][1:]),
                rec_pe=h(job["rec_pe"]), rec_npe=h(job["rec_npe"]), edit_ref=h(job["edit_ref"]), rec_ref=h(job["rec_ref"]))


def gen_ref_pipeline(name):
    import tempfile
    from oracle import ref_pipeline
    spec = REF_PIPELINE_JOBS[name]
    cfg = UNetConfig.mini() if spec["cfg"] == "mini" else UNetConfig.i2vgen_xl()
    unet = build_oracle(cfg, random_state_dict(cfg, spec["seed"]), dtype=torch.float32)
    frames, edited = ref_pipeline_frames(spec)
    with tempfile.TemporaryDirectory() as tmp:
        job = ref_pipeline.run_reference_job(unet, cfg.cross_attention_dim, frames, edited, spec["size"], spec["n_steps"],
                                             spec["ratios"], tmp)
    fx = pack_ref_pipeline_job(spec, job)
    torch.save(fx, os.path.join(HERE, spec["file"]))
    print(spec["file"], {k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in fx.items() if k != "spec"})
    print("  |edit_ref| max", float(fx["edit_ref"].float().abs().max()), " inv_ts", fx["inv_ts"])


def gen_consisti2v():
    """``consisti2v_decoder_hooks.pt`` (``--consisti2v``): the reference's own ``VideoLDMCrossAttnUpBlock``
    (``consisti2v/consisti2v/models/videoldm_unet_blocks.py:548-745``, with everything below it from the reference's files, see
    ``oracle.ref_stubs.load_reference_consisti2v_decoder``) as stand-ins for ``unet.up_blocks[1..3]``, the reference's own
    ``consisti2v/pnp_utils.py`` hooks registered on them; outputs un-hooked and hooked at t = 981 (conv + spatial + temporal
    injection) / 301 (temporal only); t = 101 (no schedule) is checked to equal the un-hooked output and not stored."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import consisti2v_spec as spec
    att, blocks_mod, ublocks, pnp = ref_stubs.load_reference_consisti2v_decoder()
    blocks = {i: spec.fill_weights(ublocks.VideoLDMCrossAttnUpBlock(**spec.block_kwargs(i))).eval() for i in spec.BLOCKS}

    def call(blk, x, skips, temb, ehs):
        with torch.no_grad():
            return blk(x, skips, temb, encoder_hidden_states=ehs).clone()
    out = spec.run_cases(blocks, pnp, call)
    fx = {"spec": dict(B=spec.B, FR=spec.FR, H=spec.H, W=spec.W, weight_seed=spec.WEIGHT_SEED, input_seed=spec.INPUT_SEED, pnp=spec.PNP)}
    for i in spec.BLOCKS:
        a, b = out[f"block{i}_nohook"], out[f"block{i}_hook_t101"]
        assert torch.equal(a, b), "a timestep outside every schedule must leave the block un-hooked"
        fx[f"block{i}_nohook"] = a
        for t in spec.TS_CASES:
            h = out[f"block{i}_hook_t{t}"]
            fx[f"block{i}_hook_t{t}"] = h
            third = h.shape[0] // 3
            print(f"block{i} t={t}: shape {tuple(h.shape)} max {float(h.abs().max()):.3f}  hooked vs un-hooked (branches 1-2) "
                  f"{float((h[third:] - a[third:]).abs().max() / a.abs().max()):.3f}  source branch unchanged "
                  f"{bool(torch.equal(h[:third], a[:third]))}")
    torch.save(fx, os.path.join(HERE, "consisti2v_decoder_hooks.pt"))


def gen_consisti2v_unet():
    """``consisti2v_unet.pt`` (``--consisti2v-unet``): the reference's own ``VideoLDMUNet3DConditionModel``
    (``consisti2v/consisti2v/models/videoldm_unet.py``, every block from the reference's files, see
    ``oracle.ref_stubs.load_reference_consisti2v_unet``) at toy width, predictions un-hooked and with the reference's own
    ``consisti2v/pnp_utils.py`` hooks registered on ``unet.up_blocks[1..3]`` at t = 981 / 301 (t = 101: checked un-hooked)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import consisti2v_spec as spec
    unet_mod, ublocks, pnp = ref_stubs.load_reference_consisti2v_unet()
    unet = spec.fill_weights(unet_mod.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)).eval()

    def call(u, sample, t, ehs, first, stride):
        with torch.no_grad():
            return u(sample, t, encoder_hidden_states=ehs, first_frame_latents=first, frame_stride=stride).sample.clone()
    out = spec.run_unet_cases(unet, pnp, call)
    a = out["unet_nohook"]
    fx = {"spec": dict(cfg=spec.UNET_CFG, H=spec.UNET_H, W=spec.UNET_W, t=spec.UNET_T, frame_stride=spec.UNET_STRIDE,
                       weight_seed=spec.WEIGHT_SEED, input_seed=spec.INPUT_SEED, pnp=spec.PNP,
                       n_params=sum(p.numel() for p in unet.parameters())), "unet_nohook": a}
    assert torch.equal(out["unet_nohook_t101"], out["unet_hook_t101"]), "a timestep outside every schedule must leave the UNet un-hooked"
    for t in spec.TS_CASES:
        h = out[f"unet_hook_t{t}"]
        fx[f"unet_hook_t{t}"] = h
        print(f"unet t={t}: shape {tuple(h.shape)} max {float(h.abs().max()):.3f}  source branch max {float(h[:1].abs().max()):.3f}  "
              f"editing vs source {float((h[2] - h[0]).abs().max()):.3f}")
    print(f"un-hooked: max {float(a.abs().max()):.3f}; hooked t=981 vs un-hooked at t=981, branches 1-2: "
          f"{float((out['unet_hook_t981'][1:] - a[1:]).abs().max() / a.abs().max()):.3f}; source branch equal "
          f"{bool(torch.equal(out['unet_hook_t981'][:1], a[:1]))}")
    torch.save(fx, os.path.join(HERE, "consisti2v_unet.pt"))


def gen_consisti2v_unet_full():
    """``consisti2v_unet_full.pt`` (``--consisti2v-unet-full``): the reference's own ``VideoLDMUNet3DConditionModel`` at the released
    model's width (1250 M parameters, fp32 on the CPU: a few minutes), un-hooked and with the reference's three hook families on."""
    import time
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import consisti2v_spec as spec
    unet_mod, ublocks, pnp = ref_stubs.load_reference_consisti2v_unet()
    t0 = time.time()
    unet = spec.fill_weights(unet_mod.VideoLDMUNet3DConditionModel(**spec.unet_full_cfg())).eval()
    print(f"reference UNet built: {sum(p.numel() for p in unet.parameters()) / 1e6:.1f} M parameters, {time.time() - t0:.0f} s")

    def call(u, sample, t, ehs, first, stride):
        t1 = time.time()
        with torch.no_grad():
            y = u(sample, t, encoder_hidden_states=ehs, first_frame_latents=first, frame_stride=stride).sample.clone()
        print(f"  forward {time.time() - t1:.0f} s, |y| max {float(y.abs().max()):.3f}")
        return y
    out = spec.run_unet_full_cases(unet, pnp, call)
    a, h = out["full_nohook"], out["full_hook_t981"]
    print(f"hooked vs un-hooked, branches 1-2: {float((h[1:] - a[1:]).abs().max() / a.abs().max()):.3f}; source branch equal {bool(torch.equal(h[:1], a[:1]))}")
    fx = {"spec": dict(cfg=spec.unet_full_cfg(), H=spec.FULL_H, W=spec.FULL_W, weight_seed=spec.WEIGHT_SEED, input_seed=spec.INPUT_SEED),
          "full_nohook": a.half(), "full_hook_t981": h.half()}
    torch.save(fx, os.path.join(HERE, "consisti2v_unet_full.pt"))


def gen_consisti2v_pipeline():
    """``consisti2v_pipeline.pt`` (``--consisti2v-pipeline``): the reference's own ``ConditionalVideoEditingPipeline`` class
    (``oracle.ref_consisti2v_pipeline``: verbatim pipeline file around the reference's UNet, hooks, inverse scheduler and
    ``load_ddim_latents_at_t``; toy VAE / text encoder) through both stages on one synthetic clip -- ``tests/consisti2v_spec.PIPE_JOB``."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import consisti2v_spec as spec
    from oracle import ref_consisti2v_pipeline as rcp
    j = spec.PIPE_JOB
    frames, edited = spec.pipeline_frames()
    with tempfile.TemporaryDirectory() as tmp:
        job = rcp.run_reference_job(spec.UNET_CFG, spec.fill_weights, frames, edited, j["height"], j["width"], j["n_inv_steps"],
                                    j["n_steps"], j["t_idx"], j["ratios"], tmp, frame_stride=j["frame_stride"],
                                    edit_prompt=j["edit_prompt"], neg=j["neg"], cfg_txt=j["cfg_txt"])
    h = lambda x: x.detach().to(torch.float16).contiguous()
    fx = dict(spec=dict(j), inv_ts=job["inv_ts"], ts=job["ts"], t0=job["t0"], lat0=h(job["lat0"]),
              trajectory=h(torch.stack([job["files"][t] for t in job["inv_ts"]])), rec_lat=h(job["rec_lat"]), edit_lat=h(job["edit_lat"]),
              edit_video=h(job["edit_video"]), rec_video=h(job["rec_video"]))
    torch.save(fx, os.path.join(HERE, "consisti2v_pipeline.pt"))
    print("consisti2v_pipeline.pt", {k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in fx.items() if k != "spec"})
    print("  |edit_lat| max", float(fx["edit_lat"].float().abs().max()), " edit vs reconstruction",
          float((fx["edit_lat"].float() - fx["rec_lat"].float()).abs().max()))


def gen_consisti2v_sampling():
    """``consisti2v_sampling.pt`` (``--consisti2v-sampling``): the reference's ``ConditionalAnimationPipeline``,
    ``AutoregressiveAnimationPipeline`` and (with ``guidance_rescale`` + ``eta``) ``ConditionalVideoEditingPipeline`` classes sampling
    from seeded noise -- ``tests/consisti2v_spec.sampling_cases``."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import consisti2v_spec as spec
    from oracle import ref_consisti2v_pipeline as rcp
    with tempfile.TemporaryDirectory() as tmp:
        got = rcp.run_reference_sampling(spec.UNET_CFG, spec.fill_weights, spec.sampling_cases(), spec.sampling_first_frame(),
                                         spec.SAMPLING_FILTER, spec.SAMPLING_SEED, tmp)
    fx = {k: v.detach().to(torch.float16).contiguous() for k, v in got.items()}
    fx["spec"] = dict(cases={k: [c, dict(kw)] for k, (c, kw) in spec.sampling_cases().items()}, seed=spec.SAMPLING_SEED,
                      filter=dict(spec.SAMPLING_FILTER))
    torch.save(fx, os.path.join(HERE, "consisti2v_sampling.pt"))
    print("consisti2v_sampling.pt", {k: tuple(v.shape) for k, v in fx.items() if k != "spec"})


def gen_seine():
    """``seine_decoder_hooks.pt`` (``--seine``): the reference's own ``CrossAttnUpBlock3D`` (``seine/models/unet_blocks.py:444-575``
    with ``seine/models/attention.py`` / ``resnet.py`` below it, ``oracle.ref_stubs.load_reference_seine_decoder``) as stand-ins
    for ``unet.up_blocks[1..3]``, the reference's own ``seine/pnp_utils.py`` hooks (conv, spatial, CROSS and temporal attention)
    registered on them; outputs un-hooked and hooked at t = 981 (all four) / 501 (cross + temporal) / 301 (temporal only)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import seine_spec as spec
    att, ublocks, res, pnp, Rotary = ref_stubs.load_reference_seine_decoder()
    rot = Rotary(32)   # seine/models/unet.py:185: ONE embedding object shared by every block
    blocks = {i: spec.fill_weights(ublocks.CrossAttnUpBlock3D(rotary_emb=rot, **spec.block_kwargs(i)), spec.WEIGHT_SEED).eval()
              for i in spec.BLOCKS}

    def call(blk, x, skips, temb, ehs):
        with torch.no_grad():
            return blk(x, skips, temb, encoder_hidden_states=ehs, use_image_num=0).clone()
    out = spec.run_cases(blocks, pnp, call)
    fx = {"spec": dict(B=spec.B, FR=spec.FR, H=spec.H, W=spec.W, weight_seed=spec.WEIGHT_SEED, input_seed=spec.INPUT_SEED, pnp=spec.PNP)}
    for i in spec.BLOCKS:
        a = out[f"block{i}_nohook"]
        assert torch.equal(a, out[f"block{i}_hook_t101"]), "a timestep outside every schedule must leave the block un-hooked"
        fx[f"block{i}_nohook"] = a
        prev = a
        for t in reversed(spec.TS_CASES):
            h = out[f"block{i}_hook_t{t}"]
            fx[f"block{i}_hook_t{t}"] = h
            print(f"block{i} t={t}: shape {tuple(h.shape)} max {float(h.abs().max()):.3f}  vs un-hooked (branches 1-2) "
                  f"{float((h[1:] - a[1:]).abs().max() / a.abs().max()):.3f}  vs the previous case "
                  f"{float((h[1:] - prev[1:]).abs().max() / a.abs().max()):.3f}  source branch unchanged {bool(torch.equal(h[:1], a[:1]))}")
            prev = h
    torch.save(fx, os.path.join(HERE, "seine_decoder_hooks.pt"))


def gen_seine_unet():
    """``seine_unet.pt`` (``--seine-unet``): the reference's own ``UNet3DConditionModel`` (``seine/models/unet.py``, every block from
    the reference's files, ``oracle.ref_stubs.load_reference_seine_decoder(with_unet=True)``) at toy width, un-hooked and with the
    reference's own four hook families registered on ``unet.up_blocks[1..3]`` (``register_time`` also walks the encoder and mid block)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import seine_spec as spec
    att, ublocks, res, pnp, Rotary = ref_stubs.load_reference_seine_decoder(with_unet=True)
    unet = spec.fill_weights(ublocks.unet.UNet3DConditionModel(**spec.UNET_CFG), spec.WEIGHT_SEED).eval()

    def call(u, sample, t, ehs):
        with torch.no_grad():
            return u(sample, t, encoder_hidden_states=ehs).sample.clone()
    out = spec.run_unet_cases(unet, pnp, call)
    assert torch.equal(out["unet_nohook_t101"], out["unet_hook_t101"]), "a timestep outside every schedule must leave the UNet un-hooked"
    a = out["unet_nohook"]
    fx = {"spec": dict(cfg=spec.UNET_CFG, H=spec.UNET_H, W=spec.UNET_W, F=spec.UNET_F, weight_seed=spec.WEIGHT_SEED, input_seed=spec.INPUT_SEED,
                       pnp=spec.PNP, n_keys=len(unet.state_dict())), "unet_nohook": a}
    for t in spec.TS_CASES:
        h = out[f"unet_hook_t{t}"]
        fx[f"unet_hook_t{t}"] = h
        print(f"seine unet t={t}: shape {tuple(h.shape)} max {float(h.abs().max()):.3f}  editing vs source {float((h[2] - h[0]).abs().max()):.3f}")
    print(f"hooked t=981 vs un-hooked, branches 1-2: {float((out['unet_hook_t981'][1:] - a[1:]).abs().max() / a.abs().max()):.3f}; "
          f"source branch equal {bool(torch.equal(out['unet_hook_t981'][:1], a[:1]))}")
    torch.save(fx, os.path.join(HERE, "seine_unet.pt"))


def gen_seine_pipeline():
    """``seine_pipeline.pt`` (``--seine-pipeline``): the reference's own runner classes (``oracle/ref_seine_pipeline.py``: both runner
    files imported verbatim around the reference's UNet, hooks and helpers; toy VAE / text encoder) through both stages on one
    synthetic clip -- ``tests/seine_spec.JOB`` -- once with the DDIM sampler, once with the shipped default, ancestral DDPM."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import seine_spec as spec
    from oracle import ref_seine_pipeline as rsp
    frames, edited = spec.job_frames()
    h = lambda x: x.detach().to(torch.float16).contiguous()
    fx = {"spec": dict(spec.JOB)}
    for sm in ("ddim", "ddpm"):
        inv, ed = spec.job_configs(sm)
        with tempfile.TemporaryDirectory() as tmp:
            job = rsp.run_reference_job(spec.UNET_CFG, spec.fill_weights, spec.WEIGHT_SEED, frames, edited, inv, ed, tmp)
        if sm == "ddim":
            fx.update(inv_ts=job["inv_ts"], lat0=h(job["lat0"]), trajectory=h(torch.stack([job["files"][t] for t in job["inv_ts"]])),
                      recon_lat=h(job["recon_lat"]), recon_frames=job["recon_frames"])
        fx[f"edit_ts_{sm}"], fx[f"edit_lat_{sm}"], fx[f"edited_frames_{sm}"] = job["edit_ts"], h(job["edit_lat"]), job["edited_frames"]
        print(sm, "edit timesteps", job["edit_ts"], "|edit_lat| max", float(job["edit_lat"].abs().max()))
    torch.save(fx, os.path.join(HERE, "seine_pipeline.pt"))
    print("seine_pipeline.pt", {k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in fx.items() if k != "spec"})


VAE_BLOCK_SPEC = dict(weight_seed=21, input_seed=22, groups=32, cases=((128, 128), (128, 256)), hw=(12, 10), n=2, sampler_c=128)


def vae_block_inputs(spec=VAE_BLOCK_SPEC):
    """Seeded inputs / fp16-rounded weights of the VAE block fixture (shared by the generator, the CPU test and the GPU check)."""
    g = torch.Generator().manual_seed(spec["input_seed"])
    H, W = spec["hw"]
    out = {"x": {}, "weights": {}}
    for cin, cout in spec["cases"]:
        out["x"][f"res{cin}_{cout}"] = torch.randn(spec["n"], cin, H, W, generator=g).half().float()
    c = spec["sampler_c"]
    out["x"]["up"] = torch.randn(spec["n"], c, H, W, generator=g).half().float()
    xd = torch.randn(spec["n"], c, H, W, generator=g).half().float()
    xd[:, :, 0, :] = 0   # first row / column zero: F.pad(x, (0, 1, 0, 1)) then equals the symmetric pad of x[1:, 1:] (see gen_vae_blocks)
    xd[:, :, :, 0] = 0
    out["x"]["down"] = xd
    gw = torch.Generator().manual_seed(spec["weight_seed"])

    def w(*shape, scale):
        return (torch.randn(*shape, generator=gw) * scale).half().float()
    for cin, cout in spec["cases"]:
        sd = {"norm1.weight": 1 + w(cin, scale=0.1), "norm1.bias": w(cin, scale=0.05),
              "conv1.weight": w(cout, cin, 3, 3, scale=(9 * cin) ** -0.5), "conv1.bias": w(cout, scale=0.05),
              "norm2.weight": 1 + w(cout, scale=0.1), "norm2.bias": w(cout, scale=0.05),
              "conv2.weight": w(cout, cout, 3, 3, scale=(9 * cout) ** -0.5), "conv2.bias": w(cout, scale=0.05)}
        if cin != cout:
            sd["conv_shortcut.weight"] = w(cout, cin, 1, 1, scale=cin ** -0.5)
            sd["conv_shortcut.bias"] = w(cout, scale=0.05)
        out["weights"][f"res{cin}_{cout}"] = sd
    for name in ("up", "down"):
        out["weights"][name] = {"conv.weight": w(c, c, 3, 3, scale=(9 * c) ** -0.5), "conv.bias": w(c, scale=0.05)}
    return out


def gen_vae_blocks():
    """``vae_blocks_ref.pt`` (``--vae-blocks``): BLOCK-level pin of ``oracle/vae_oracle.py``.  No ``AutoencoderKL``
    source exists under the reference tree, but its building blocks are diffusers' ResNet block / samplers, which the reference
    vendors at ``seine/models/resnet.py``: ``ResnetBlock3D`` (``:113-207``) with ``temb_channels=None``, ``eps=1e-6`` at ONE frame ==
    the VAE's ResnetBlock2D(temb=None); ``Upsample3D(use_conv=True)`` (``:24-76``) == nearest x2 + 3x3 conv; ``Downsample3D``
    (``:79-110``) is the stride-2 3x3 conv with SYMMETRIC padding 1 (its ``padding == 0`` branch -- the VAE's -- raises
    NotImplementedError there), so the asymmetric form ``conv(F.pad(x, (0, 1, 0, 1)), stride 2, pad 0)`` is pinned through the
    identity ``F.pad(x, (0, 1, 0, 1)) == F.pad(x[..., 1:, 1:], (1, 1, 1, 1))`` for an x whose first row and column are zero: the
    reference class runs on ``x[..., 1:, 1:]``.  The encoder / decoder ASSEMBLY and the mid-block attention stay unpinnable."""
    res, _ = ref_stubs.load_reference_seine_blocks()
    spec = VAE_BLOCK_SPEC
    io = vae_block_inputs(spec)
    fx = {"spec": dict(spec), "out": {}}
    with torch.no_grad():
        for cin, cout in spec["cases"]:
            name = f"res{cin}_{cout}"
            blk = res.ResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=None, groups=spec["groups"], eps=1e-6).eval()
            blk.load_state_dict(io["weights"][name])
            fx["out"][name] = blk(io["x"][name][:, :, None], None)[:, :, 0].clone()
        c = spec["sampler_c"]
        up = res.Upsample3D(c, use_conv=True).eval()
        up.load_state_dict(io["weights"]["up"])
        fx["out"]["up"] = up(io["x"]["up"][:, :, None])[:, :, 0].clone()
        dn = res.Downsample3D(c, use_conv=True, padding=1).eval()
        dn.conv.load_state_dict({k[len("conv."):]: v for k, v in io["weights"]["down"].items()})
        fx["out"]["down"] = dn(io["x"]["down"][:, :, None, 1:, 1:])[:, :, 0].clone()
    torch.save(fx, os.path.join(HERE, "vae_blocks_ref.pt"))
    print("vae_blocks_ref.pt", {k: tuple(v.shape) for k, v in fx["out"].items()})


if __name__ == "__main__":
    assert ref_stubs.reference_available(), "needs /root/reference"
    if "--vae-blocks" in sys.argv:
        gen_vae_blocks()
        sys.exit(0)
    if "--seine" in sys.argv:
        gen_seine()
        sys.exit(0)
    if "--seine-pipeline" in sys.argv:
        gen_seine_pipeline()
        sys.exit(0)
    if "--seine-unet" in sys.argv:
        gen_seine_unet()
        sys.exit(0)
    if "--consisti2v" in sys.argv:
        gen_consisti2v()
    if "--consisti2v-unet" in sys.argv:
        gen_consisti2v_unet()
        sys.exit(0)
    if "--consisti2v-unet-full" in sys.argv:
        gen_consisti2v_unet_full()
        sys.exit(0)
    if "--consisti2v-pipeline" in sys.argv:
        gen_consisti2v_pipeline()
    if "--consisti2v-sampling" in sys.argv:
        gen_consisti2v_sampling()
        sys.exit(0)
    if "--pipeline" in sys.argv:
        gen_ref_pipeline("mini")
    elif "--pipeline-full" in sys.argv:
        gen_ref_pipeline("full")
    elif "--full" in sys.argv:
        gen_hooks_full()
    else:
        gen_hooks()
        gen_scheduler()
