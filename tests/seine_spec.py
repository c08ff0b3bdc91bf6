"""Shared definition of the SEINE decoder-hook fixture (tests/golden/seine_decoder_hooks.pt): see consisti2v_spec.py -- weights and
inputs are re-derived from names and seeds by the generator (the REFERENCE's own ``CrossAttnUpBlock3D`` + ``seine/pnp_utils.py``,
``tests/golden/make_golden.py --seine``) and by the tests (native blocks); the fixture holds outputs only."""
import types

import torch
from torch import nn

from consisti2v_spec import B, CROSS, FR, GROUPS, H, INPUT_SEED, N_STEPS, TEMB, TOKENS, W, fill_weights  # noqa: F401 (same geometry)

PNP = dict(pnp_f_t=0.2, pnp_spatial_attn_t=0.4, pnp_cross_attn_t=0.6, pnp_temp_attn_t=0.8)
TS_CASES = (981, 501, 301)        # every hook injecting / cross + temporal / temporal only
WEIGHT_SEED = 9753

BLOCKS = {   # stand-ins for unet.up_blocks[1..3]; attn_num_head_channels = number of heads (seine/models/unet_blocks.py:498-500)
    1: dict(in_channels=64, out_channels=128, prev_output_channel=128, attn_num_head_channels=2, add_upsample=True),
    2: dict(in_channels=64, out_channels=64, prev_output_channel=128, attn_num_head_channels=2, add_upsample=True),
    3: dict(in_channels=64, out_channels=64, prev_output_channel=64, attn_num_head_channels=1, add_upsample=False),
}


def block_kwargs(i):
    return dict(temb_channels=TEMB, num_layers=3, resnet_eps=1e-5, resnet_groups=GROUPS, cross_attention_dim=CROSS,
                use_linear_projection=(i == 2), use_first_frame=False, use_relative_position=False, **BLOCKS[i])


def block_inputs(i, seed=INPUT_SEED):
    k = BLOCKS[i]
    g = torch.Generator().manual_seed(seed + 100 + i)
    cout, cin, prev = k["out_channels"], k["in_channels"], k["prev_output_channel"]
    r = lambda *s: torch.randn(*s, generator=g).half().float()
    x = r(B, prev, FR, H, W)
    skips = (r(B, cin, FR, H, W), r(B, cout, FR, H, W), r(B, cout, FR, H, W))
    return x, skips, r(B, TEMB), r(B, TOKENS, CROSS)


def schedules():
    ts = torch.arange(N_STEPS).flip(0) * (1000 // N_STEPS) + 1
    return {k: ts[: int(N_STEPS * v)] for k, v in PNP.items()}


class StubUNet(nn.Module):
    """``model.unet`` as ``seine/pnp_utils.py`` walks it: ``up_blocks[1..3]`` are the blocks under test; ``down_blocks`` / ``mid_block``
    only receive ``register_time``'s attribute writes (``:133-147``) and hold plain namespaces."""

    def __init__(self, blocks):
        super().__init__()
        self.up_blocks = nn.ModuleList([nn.Identity(), blocks[1], blocks[2], blocks[3]])
        site = lambda: types.SimpleNamespace(transformer_blocks=[types.SimpleNamespace(
            attn1=types.SimpleNamespace(), attn2=types.SimpleNamespace(), attn_temp=types.SimpleNamespace())])
        self.__dict__["down_blocks"] = [types.SimpleNamespace(attentions=[site(), site()]) for _ in range(3)]
        self.__dict__["mid_block"] = types.SimpleNamespace(attentions=[site()])


def run_cases(blocks, pnp_module, call):
    out = {}
    for i, blk in blocks.items():
        out[f"block{i}_nohook"] = call(blk, *block_inputs(i))
    model = types.SimpleNamespace(unet=StubUNet(blocks))
    s = schedules()
    pnp_module.register_conv_injection(model, s["pnp_f_t"])
    pnp_module.register_spatial_attention_pnp(model, s["pnp_spatial_attn_t"])
    pnp_module.register_cross_attention_pnp(model, s["pnp_cross_attn_t"])
    pnp_module.register_temp_attention_pnp(model, s["pnp_temp_attn_t"])
    for t in TS_CASES + (101,):
        pnp_module.register_time(model, t)
        for i, blk in blocks.items():
            out[f"block{i}_hook_t{t}"] = call(blk, *block_inputs(i))
    return out


# ------------------------------------------------------------------------------------------------- the whole UNet (seine_unet.pt)
# UNet3DConditionModel with the released model's options at toy width: 4 levels, 2 layers per block, 9 input channels
# (latents | mask | masked-video latents), 1 x 1-conv projections, head_dim >= 32 everywhere (RotaryEmbedding(32) per head).
UNET_CFG = dict(sample_size=8, in_channels=9, out_channels=4, block_out_channels=(64, 64, 128, 128), layers_per_block=2, norm_num_groups=8,
                cross_attention_dim=CROSS, attention_head_dim=(2, 1, 2, 4), use_linear_projection=False)
UNET_H, UNET_W, UNET_F = 8, 16, 4


def unet_inputs(seed=INPUT_SEED):
    g = torch.Generator().manual_seed(seed + 277)
    r = lambda *s: torch.randn(*s, generator=g).half().float()
    return r(B, 9, UNET_F, UNET_H, UNET_W), r(B, TOKENS, CROSS)


def run_unet_cases(unet, pnp_module, call):
    """``call(unet, sample, t, ehs)`` -> prediction; un-hooked (t = 981, 101), then with the FOUR hook families at each TS_CASES (+ 101)."""
    out = {}
    sample, ehs = unet_inputs()
    out["unet_nohook"] = call(unet, sample, 981, ehs)
    out["unet_nohook_t101"] = call(unet, sample, 101, ehs)
    model = types.SimpleNamespace(unet=unet)
    s = schedules()
    pnp_module.register_conv_injection(model, s["pnp_f_t"])
    pnp_module.register_spatial_attention_pnp(model, s["pnp_spatial_attn_t"])
    pnp_module.register_cross_attention_pnp(model, s["pnp_cross_attn_t"])
    pnp_module.register_temp_attention_pnp(model, s["pnp_temp_attn_t"])
    for t in TS_CASES + (101,):
        pnp_module.register_time(model, t)
        out[f"unet_hook_t{t}"] = call(unet, sample, t, ehs)
    return out


# ------------------------------------------------------------------------------------------------- the two-stage job (seine_pipeline.pt)
JOB = dict(frames=UNET_F, height=64, width=128, inv_steps=8, save_steps=4, edit_steps=4, cfg_scale=4.0, prompt="a robot", negative_prompt="blurry",
           pnp=dict(pnp_f_t=0.5, pnp_spatial_attn_t=0.5, pnp_cross_attn_t=0.25, pnp_temp_attn_t=0.75))


def job_configs(sample_method="ddpm"):
    """The reference's two yaml files (``configs/seine/``: same keys and values as ``seine/configs/``) with the toy job's overrides."""
    import os
    from anyv2v_amd.config import OmegaConf
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "seine")
    inv, ed = OmegaConf.load(os.path.join(root, "ddim_inversion.yaml")), OmegaConf.load(os.path.join(root, "pnp_edit.yaml"))
    j = JOB
    inv.device = ed.device = "cpu"
    inv.image_size = ed.image_size = [j["height"], j["width"]]
    inv.n_steps, inv.n_save_steps, inv.n_frame_to_invert = j["inv_steps"], j["save_steps"], j["frames"]
    ed.n_ddim_inversion_steps, ed.n_frame_inverted, ed.n_frames, ed.n_steps = j["inv_steps"], j["frames"], j["frames"], j["edit_steps"]
    ed.sample_method, ed.cfg_scale, ed.prompt, ed.negative_prompt = sample_method, j["cfg_scale"], j["prompt"], j["negative_prompt"]
    for k, v in j["pnp"].items():
        ed[k] = v
    return inv, ed


def job_frames():
    import numpy as np
    from PIL import Image
    j = JOB
    rng = np.random.RandomState(7)

    def pic(ph):
        yy, xx = np.mgrid[0:j["height"], 0:j["width"]].astype(np.float32)
        yy, xx = yy / j["height"], xx / j["width"]
        tex = rng.rand(j["height"], j["width"], 3).astype(np.float32)
        img = np.stack([0.5 + 0.5 * np.sin(6.3 * (xx + ph)), yy, 0.5 + 0.5 * np.cos(6.3 * (xx * yy + ph))], -1)
        return Image.fromarray((255 * (0.8 * img + 0.2 * tex)).clip(0, 255).astype("uint8"))
    return [pic(i / 3.0) for i in range(j["frames"])], pic(0.41).transpose(Image.FLIP_LEFT_RIGHT)


def native_job(device, work_dir, sample_method="ddpm", trajectory_from=None):
    """Both stages on the native runner classes, driven like ``oracle.ref_seine_pipeline.run_reference_job`` drives the reference's.
    ``trajectory_from``: {t: latents} written in place of the native inversion's own files before the edit stage."""
    import os
    from anyv2v_amd.seine_pipeline import SEINEDDIMInversionPipeline, SEINEPnPPipeline
    from anyv2v_amd import seine as sn
    from anyv2v_amd.schedulers import SEINE_SCHEDULER_CONFIG, DDIMScheduler
    from anyv2v_amd.utils import seed_everything
    from consisti2v_spec import ToyVaeAdapter
    from hf_clip_reference import HFTextEncoder
    from oracle import ref_consisti2v_pipeline as rcp
    from oracle import ref_pipeline as rp
    from pathlib import Path
    import yaml
    work_dir = str(work_dir)
    frames, edited = job_frames()
    clip_dir = os.path.join(work_dir, "clip")
    os.makedirs(clip_dir, exist_ok=True)
    for i, f in enumerate(frames):
        f.save(os.path.join(clip_dir, "%05d.png" % i))
    edited_path = os.path.join(work_dir, "edited.png")
    edited.save(edited_path)
    inv, ed = job_configs(sample_method)
    inv.device = ed.device = str(device)
    dim = UNET_CFG["cross_attention_dim"]

    def parts():
        unet = fill_weights(sn.UNet3DConditionModel(**UNET_CFG), WEIGHT_SEED).to(device)
        tok = rp.ToyTokenizer()
        return dict(unet=unet, vae=ToyVaeAdapter(rcp.ToyVAE()), text_encoder=HFTextEncoder(rp.ToyTextEncoder(dim), tok))
    inv.src_video_path, inv.output_dir = clip_dir, os.path.join(work_dir, "ddim-inversion", "default")
    seed_everything(inv.seed)
    toy = DDIMScheduler(**SEINE_SCHEDULER_CONFIG)
    toy.set_timesteps(inv.n_save_steps)
    save_path = os.path.join(inv.output_dir, inv.model_name, Path(inv.src_video_path).stem, f"steps_{inv.n_steps}", f"nframes_{inv.n_frame_to_invert}")
    os.makedirs(os.path.join(save_path, "ddim_latents"), exist_ok=True)
    with open(os.path.join(save_path, "inversion_prompts.yaml"), "w") as f:
        yaml.dump({Path(inv.src_video_path).stem: inv.inversion_prompt}, f)
    p1 = SEINEDDIMInversionPipeline(device, inv, **parts())
    out = {"lat0": p1.latent_at_0.clone()}
    out["recon_frames"] = p1.extract_ddim_latents(inv, toy.timesteps, save_path)
    out["recon_lat"] = p1.reconstructed_latents
    lat_dir = os.path.join(save_path, "ddim_latents")
    out["files"] = {int(f.split("_")[-1].split(".")[0]): torch.load(os.path.join(lat_dir, f)) for f in sorted(os.listdir(lat_dir))}
    if trajectory_from is not None:
        for t, v in trajectory_from.items():
            torch.save(v.clone(), os.path.join(lat_dir, f"ddim_latents_{t}.pt"))
    ed.src_video_path, ed.edited_first_frame_path, ed.ddim_inversion_dir = clip_dir + ".mp4", edited_path, inv.output_dir
    comp = parts()              # (built before seeding: the DDPM noise stream is then a function of the seed alone)
    seed_everything(ed.seed)
    p2 = SEINEPnPPipeline(device, ed, **comp)
    p2.scheduler.noise_on_host = True     # the fixture's noise stream is the CPU generator's
    p2.scheduler.set_timesteps(ed.n_steps)
    if ed.enable_pnp:
        p2.init_pnp()
    out["edited_frames"] = p2.edit_video(ed)
    out["edit_lat"] = p2.edited_latents
    out["edit_ts"] = [int(t) for t in p2.scheduler.timesteps]
    out["pipe"] = p2
    return out
