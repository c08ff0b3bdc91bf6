"""Shared definition of the ConsistI2V decoder-hook fixture (tests/golden/consisti2v_decoder_hooks.pt): block shapes, weights and
inputs are re-derived from names and seeds on both sides -- ``tests/golden/make_golden.py --consisti2v`` (the REFERENCE's own
``VideoLDMCrossAttnUpBlock`` + ``consisti2v/pnp_utils.py``, CPU fp32) and the tests (native blocks, op emulation on the CPU, HIP
kernels on the GPU) -- so the fixture holds outputs only."""
import types
import zlib

import torch

B, FR, H, W = 3, 4, 3, 4          # [source, negative, editing] x 4 frames, 3 x 4 latent pixels (not square on purpose)
TEMB, CROSS, TOKENS, GROUPS = 64, 48, 7, 8
N_STEPS = 50
PNP = dict(pnp_f_t=0.2, pnp_spatial_attn_t=0.5, pnp_temp_attn_t=0.8)
TS_CASES = (981, 301)             # every site injecting / temporal attention only
WEIGHT_SEED, INPUT_SEED = 4321, 8888

# stand-ins for unet.up_blocks[1..3] (three layers each, as the reference builds them: layers_per_block + 1)
BLOCKS = {
    1: dict(in_channels=64, out_channels=128, prev_output_channel=128, num_attention_heads=2, n_temp_heads=4, add_upsample=True),
    2: dict(in_channels=64, out_channels=64, prev_output_channel=128, num_attention_heads=1, n_temp_heads=4, add_upsample=True),
    3: dict(in_channels=64, out_channels=64, prev_output_channel=64, num_attention_heads=1, n_temp_heads=1, add_upsample=False),
}


def block_kwargs(i):
    return dict(temb_channels=TEMB, num_layers=3, resnet_eps=1e-5, resnet_groups=GROUPS, cross_attention_dim=CROSS,
                use_linear_projection=True, use_temporal=True, augment_temporal_attention=True, n_frames=FR,
                first_frame_condition_mode="concat", rotary_emb=True, **BLOCKS[i])


def fill_weights(module, seed=WEIGHT_SEED):
    """Deterministic parameters by NAME (both module trees have the reference's state-dict keys)."""
    sd = module.state_dict()
    new = {}
    for name in sorted(sd):
        v = sd[name]
        g = torch.Generator().manual_seed(seed + zlib.crc32(name.encode()) % 100000)
        if name.endswith("freqs"):
            new[name] = v.clone()
        elif name.endswith("alpha"):
            new[name] = torch.rand(1, generator=g) * 0.6 + 0.2
        elif v.dim() >= 2:
            fan_in = v[0].numel()
            new[name] = torch.randn(v.shape, generator=g) * (1.0 / fan_in ** 0.5)
        elif "norm" in name and name.endswith("weight"):
            new[name] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            new[name] = 0.1 * torch.randn(v.shape, generator=g)
    module.load_state_dict(new)
    return module


def block_inputs(i, seed=INPUT_SEED):
    k = BLOCKS[i]
    g = torch.Generator().manual_seed(seed + i)
    N = B * FR
    cout, cin, prev = k["out_channels"], k["in_channels"], k["prev_output_channel"]
    r = lambda *s: torch.randn(*s, generator=g).half().float()     # the tests feed fp16 tensors: generate fp16-representable values
    x = r(N, prev, H, W)
    skips = (r(N, cin, H, W), r(N, cout, H, W), r(N, cout, H, W))  # consumed from the end
    temb = r(B, TEMB).repeat_interleave(FR, 0)
    ehs = r(B, TOKENS, CROSS).repeat_interleave(FR, 0)
    return x, skips, temb, ehs


def schedules():
    ts = torch.arange(N_STEPS).flip(0) * (1000 // N_STEPS) + 1
    return (ts[: int(N_STEPS * PNP["pnp_f_t"])], ts[: int(N_STEPS * PNP["pnp_spatial_attn_t"])],
            ts[: int(N_STEPS * PNP["pnp_temp_attn_t"])])


def stub_model(blocks):
    """``model.unet.up_blocks[i]`` as the hook functions index it (``consisti2v/pnp_utils.py:20-28``); index 0 holds no hook site."""
    return types.SimpleNamespace(unet=types.SimpleNamespace(up_blocks=[None, blocks[1], blocks[2], blocks[3]]))


def run_cases(blocks, pnp_module, call, log=None):
    """Outputs of every block: un-hooked, then with the hook family registered at each timestep of TS_CASES."""
    out = {}
    for i, blk in blocks.items():
        out[f"block{i}_nohook"] = call(blk, *block_inputs(i))
    model = stub_model(blocks)
    conv_s, spa_s, tmp_s = schedules()
    pnp_module.register_conv_injection(model, conv_s)
    pnp_module.register_spatial_attention_pnp(model, spa_s)
    pnp_module.register_temp_attention_pnp(model, tmp_s)
    for t in TS_CASES + (101,):
        pnp_module.register_time(model, t)
        for i, blk in blocks.items():
            out[f"block{i}_hook_t{t}"] = call(blk, *block_inputs(i))
    return out


# ------------------------------------------------------------------------------------------------- the whole UNet (consisti2v_unet.pt)
# VideoLDMUNet3DConditionModel with the released model's options at toy width: 4 levels (the hooks index up_blocks[1..3]), 2 layers
# per block, first-frame conditioning by concatenation, augmented rotary temporal attention, frame-stride conditioning.
UNET_CFG = dict(sample_size=8, in_channels=4, out_channels=4, block_out_channels=(32, 64, 64, 64), layers_per_block=2, norm_num_groups=8,
                cross_attention_dim=48, attention_head_dim=(2, 4, 4, 4), use_linear_projection=True, use_temporal=True, n_frames=4,
                n_temp_heads=2, first_frame_condition_mode="concat", augment_temporal_attention=True, temp_pos_embedding="rotary",
                use_frame_stride_condition=True)
UNET_H, UNET_W, UNET_T, UNET_STRIDE = 8, 16, 981, 3


def unet_inputs(seed=INPUT_SEED):
    """[source, negative, editing] x (n_frames - 1) noisy frames, the clean first-frame latent, text tokens."""
    g = torch.Generator().manual_seed(seed + 77)
    r = lambda *s: torch.randn(*s, generator=g).half().float()
    nf = UNET_CFG["n_frames"] - 1
    sample = r(B, UNET_CFG["in_channels"], nf, UNET_H, UNET_W)
    first = r(1, UNET_CFG["in_channels"], 1, UNET_H, UNET_W).repeat(B, 1, 1, 1, 1)   # the same clip's first frame in every branch
    ehs = r(B, TOKENS, UNET_CFG["cross_attention_dim"])
    return sample, first, ehs


def run_unet_cases(unet, pnp_module, call):
    """``call(unet, sample, t, ehs, first, frame_stride)`` -> prediction; un-hooked, then hooked at each timestep of TS_CASES (+ 101)."""
    out = {}
    sample, first, ehs = unet_inputs()
    out["unet_nohook"] = call(unet, sample, UNET_T, ehs, first, UNET_STRIDE)
    out["unet_nohook_t101"] = call(unet, sample, 101, ehs, first, UNET_STRIDE)
    model = types.SimpleNamespace(unet=unet)
    conv_s, spa_s, tmp_s = schedules()
    pnp_module.register_conv_injection(model, conv_s)
    pnp_module.register_spatial_attention_pnp(model, spa_s)
    pnp_module.register_temp_attention_pnp(model, tmp_s)
    for t in TS_CASES + (101,):
        pnp_module.register_time(model, t)
        out[f"unet_hook_t{t}"] = call(unet, sample, t, ehs, first, UNET_STRIDE)
    return out


# ------------------------------------------------------------------------------------------------- the released width (consisti2v_unet_full.pt)
# 1250 M parameters (anyv2v_amd.consisti2v_pipeline.CONSISTI2V_UNET_CONFIG), [source, negative, editing] x 15 + 1 frames x 32 x 32
# latent pixels (configs/pipeline_256): spatial head_dim 64 with Sk = 2 HW, temporal head_dim 40 / 80 / 160 with 16 + 8 keys.
FULL_H = FULL_W = 32


def unet_full_cfg():
    from anyv2v_amd.consisti2v_pipeline import CONSISTI2V_UNET_CONFIG
    return dict(CONSISTI2V_UNET_CONFIG)


def unet_full_inputs(seed=INPUT_SEED):
    cfg = unet_full_cfg()
    g = torch.Generator().manual_seed(seed + 177)
    r = lambda *s: torch.randn(*s, generator=g).half().float()
    sample = r(B, 4, cfg["n_frames"] - 1, FULL_H, FULL_W)
    first = r(1, 4, 1, FULL_H, FULL_W).repeat(B, 1, 1, 1, 1)
    ehs = r(B, 77, cfg["cross_attention_dim"])
    return sample, first, ehs


def run_unet_full_cases(unet, pnp_module, call):
    """Un-hooked and with all three hook families on (t = 981), frame stride 3."""
    sample, first, ehs = unet_full_inputs()
    out = {"full_nohook": call(unet, sample, 981, ehs, first, UNET_STRIDE)}
    model = types.SimpleNamespace(unet=unet)
    conv_s, spa_s, tmp_s = schedules()
    pnp_module.register_conv_injection(model, conv_s)
    pnp_module.register_spatial_attention_pnp(model, spa_s)
    pnp_module.register_temp_attention_pnp(model, tmp_s)
    pnp_module.register_time(model, 981)
    out["full_hook_t981"] = call(unet, sample, 981, ehs, first, UNET_STRIDE)
    return out


# ------------------------------------------------------------------------------------------------- the pipeline job (consisti2v_pipeline.pt)
# one synthetic clip through both stages as the reference's runners drive them (run_ddim_inversion.py / run_pnp_edit.py) at toy size
PIPE_JOB = dict(frames=UNET_CFG["n_frames"], height=64, width=128, seed=99, n_inv_steps=8, n_steps=4, t_idx=1, ratios=(0.5, 0.5, 0.75),
                frame_stride=3, cfg_txt=35.0, edit_prompt="a robot", neg="blurry")


def pipeline_frames():
    """Source frames (wider than the target: the centre crop matters) and an edited first frame at the target size, as PIL images."""
    import numpy as np
    from PIL import Image
    j = PIPE_JOB
    n, H, W = j["frames"], j["height"], j["width"]
    rng = np.random.RandomState(j["seed"])

    def pic(h, w, ph):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        yy, xx = yy / h, xx / w
        tex = rng.rand(h, w, 3).astype(np.float32)
        img = np.stack([0.5 + 0.5 * np.sin(6.3 * (xx + ph)), yy, 0.5 + 0.5 * np.cos(6.3 * (xx * yy + ph))], -1)
        return Image.fromarray((255 * (0.8 * img + 0.2 * tex)).clip(0, 255).astype("uint8"))
    frames = [pic(H + 16, W + 40, i / max(n - 1, 1)) for i in range(n)]
    edited = pic(H, W, 0.37).transpose(Image.FLIP_LEFT_RIGHT)
    return frames, edited


class ToyVaeAdapter:
    """``oracle.ref_consisti2v_pipeline.ToyVAE`` behind ``anyv2v_amd.encoders``' VAE interface (what NativeVAE does around the real
    AutoencoderKL)."""

    def __init__(self, toy):
        self.toy, self.config = toy, toy.config

    def to(self, device):
        return self

    def encode_pixels(self, x, device):
        z = self.toy.encode(x.float().cpu()).latent_dist.sample() * self.config.scaling_factor
        return z.detach().to(device=device, dtype=torch.float16)

    def decode_video(self, latents, decode_chunk_size=None):
        z = latents[0].permute(1, 0, 2, 3).float().cpu() / self.config.scaling_factor
        return self.toy.decode(z).sample.permute(1, 0, 2, 3)[None].float()


def native_pipeline_job(device, trajectory_from=None, work_dir=None):
    """Both stages on the native pipeline, driven like ``run_reference_job`` drives the reference's: returns the same keys.
    ``trajectory_from``: {t: latents} to edit from (the reference's files) instead of the native inversion's own."""
    from anyv2v_amd import consisti2v as c2
    from anyv2v_amd.consisti2v_pipeline import ConditionalVideoEditingPipeline
    from anyv2v_amd.schedulers import CONSISTI2V_SCHEDULER_CONFIG, DDIMInverseScheduler, DDIMScheduler
    from anyv2v_amd.utils import LatentTrajectory
    from hf_clip_reference import HFTextEncoder
    from oracle import ref_consisti2v_pipeline as rcp
    from oracle import ref_pipeline as rp
    j = PIPE_JOB
    H, W, n = j["height"], j["width"], j["frames"]
    frames, edited = pipeline_frames()
    dim = UNET_CFG["cross_attention_dim"]
    unet = fill_weights(c2.VideoLDMUNet3DConditionModel(**UNET_CFG)).to(device)
    tok = rp.ToyTokenizer()
    pipe = ConditionalVideoEditingPipeline(vae=ToyVaeAdapter(rcp.ToyVAE()), text_encoder=HFTextEncoder(rp.ToyTextEncoder(dim), tok),
                                           tokenizer=tok, unet=unet, scheduler=DDIMInverseScheduler(**CONSISTI2V_SCHEDULER_CONFIG))
    pipe._device = torch.device(device)
    out = {}
    out["lat0"] = pipe.encode_vae_video(frames, pipe.device, height=H, width=W)
    traj = pipe.invert(prompt="", first_frame_paths=frames[0], height=H, width=W, video_length=n, num_inference_steps=j["n_inv_steps"],
                       guidance_scale_txt=1.0, guidance_scale_img=1.0, negative_prompt="", frame_stride=j["frame_stride"],
                       latents=out["lat0"], output_type="latent", return_trajectory=True)
    out["inv_ts"] = [int(t) for t in pipe.scheduler.timesteps]
    out["files"] = {t: traj[t] for t in out["inv_ts"]}
    if trajectory_from is not None:
        traj = LatentTrajectory()
        for t, v in trajectory_from.items():
            traj[t] = v.to(device=device, dtype=torch.float16)
    sched = DDIMScheduler(**CONSISTI2V_SCHEDULER_CONFIG)
    sched.set_timesteps(j["n_steps"])
    ts = sched.timesteps.clone()
    t0 = int(ts[j["t_idx"]])
    pipe.register_modules(scheduler=sched)
    out["rec_lat"] = pipe(prompt="", first_frame_paths=frames[0], height=H, width=W, video_length=n, num_inference_steps=j["n_steps"],
                          guidance_scale_txt=1.0, guidance_scale_img=1.0, negative_prompt="", frame_stride=j["frame_stride"],
                          latents=traj[t0].clone(), ddim_init_latents_t_idx=j["t_idx"], output_type="latent").videos
    k = lambda r: ts[: int(j["n_steps"] * r)]
    c2.register_conv_injection(pipe, k(j["ratios"][0]))
    c2.register_spatial_attention_pnp(pipe, k(j["ratios"][1]))
    c2.register_temp_attention_pnp(pipe, k(j["ratios"][2]))
    common = dict(prompt=j["edit_prompt"], first_frame_paths=edited, height=H, width=W, video_length=n, num_inference_steps=j["n_steps"],
                  guidance_scale_txt=j["cfg_txt"], guidance_scale_img=1.0, negative_prompt=j["neg"], frame_stride=j["frame_stride"],
                  ddim_init_latents_t_idx=j["t_idx"], ddim_inv_latents_path=traj, ddim_inv_prompt="", ddim_inv_1st_frame_path=frames[0])
    out["edit_lat"] = pipe.sample_with_pnp(latents=traj[t0].clone(), output_type="latent", **common).videos
    out["edit_video"] = pipe.sample_with_pnp(latents=traj[t0].clone(), output_type="tensor", **common).videos
    out["pipe"] = pipe
    return out


# ------------------------------------------------------------------------------------------------- sampling cases (consisti2v_sampling.pt)
# the samplers / options next to the two runner stages: both animation pipelines and guidance_rescale + eta, from seeded noise
SAMPLING_SEED = 5


def sampling_cases():
    """{name: (pipeline class name, call kwargs without the first frame / generator)}."""
    j = PIPE_JOB
    base = dict(prompt="a robot", height=j["height"], width=j["width"], video_length=j["frames"], num_inference_steps=2, negative_prompt="blurry",
                frame_stride=3)
    anim = dict(base, guidance_scale_txt=3.0, guidance_scale_img=1.5, noise_sampling_method="pyoco_progressive", noise_alpha=0.7,
                use_frameinit=True, frameinit_noise_level=900)
    return {"animation": ("ConditionalAnimationPipeline", anim),
            "autoregressive": ("AutoregressiveAnimationPipeline", dict(anim, autoregress_steps=2)),
            "rescale_eta": ("ConditionalVideoEditingPipeline", dict(base, num_inference_steps=3, guidance_scale_txt=6.0, guidance_scale_img=1.0,
                                                                    guidance_rescale=0.7, eta=0.5))}


SAMPLING_FILTER = dict(method="butterworth", n=4, d_s=0.25, d_t=0.25)


def sampling_first_frame():
    """A first frame of another aspect ratio than the target (the animation pipelines resize the short side and centre-crop)."""
    from PIL import Image
    j = PIPE_JOB
    frames, _ = pipeline_frames()
    return frames[0].resize((j["width"] + 40, j["height"] + 8), resample=Image.BICUBIC)


def native_sampling(device, names=None):
    """Every sampling case on the native pipelines -> {name: latents [1, 4, F', h, w]}."""
    import types
    from anyv2v_amd import consisti2v as c2
    from anyv2v_amd import consisti2v_pipeline as cp
    from anyv2v_amd.schedulers import CONSISTI2V_SCHEDULER_CONFIG, DDIMScheduler
    from hf_clip_reference import HFTextEncoder
    from oracle import ref_consisti2v_pipeline as rcp
    from oracle import ref_pipeline as rp
    j = PIPE_JOB
    dim = UNET_CFG["cross_attention_dim"]
    unet = fill_weights(c2.VideoLDMUNet3DConditionModel(**UNET_CFG)).to(device)
    tok = rp.ToyTokenizer()
    first = sampling_first_frame()
    out = {}
    for name, (cls, kw) in sampling_cases().items():
        if names is not None and name not in names:
            continue
        pipe = getattr(cp, cls)(vae=ToyVaeAdapter(rcp.ToyVAE()), text_encoder=HFTextEncoder(rp.ToyTextEncoder(dim), tok), tokenizer=tok,
                                unet=unet, scheduler=DDIMScheduler(**CONSISTI2V_SCHEDULER_CONFIG))
        pipe._device = torch.device(device)
        if kw.get("use_frameinit"):
            pipe.init_filter(j["frames"], j["height"], j["width"], types.SimpleNamespace(**SAMPLING_FILTER))
        out[name] = pipe(first_frame_paths=first, generator=torch.Generator().manual_seed(SAMPLING_SEED), output_type="latent", **kw).videos
    return out
