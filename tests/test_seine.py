"""SURVEY.md 8(f) F4 -- the SEINE hook family (``anyv2v_amd/seine.py``) on the CPU: native decoder blocks through the op emulation
vs the fixture produced by the REFERENCE's own ``CrossAttnUpBlock3D`` + ``seine/pnp_utils.py`` (``make_golden.py --seine``), and --
where /root/reference exists -- the fixture vs the reference run live."""
import os
import types
import warnings

import numpy as np
import pytest
import torch

import cpu_ops_emulation as emu
import seine_spec as spec
from oracle import ref_stubs

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "seine_decoder_hooks.pt")


def _native_blocks():
    from anyv2v_amd import seine as sn
    return sn, {i: spec.fill_weights(sn.CrossAttnUpBlock3D(**spec.block_kwargs(i)), spec.WEIGHT_SEED) for i in spec.BLOCKS}


def _native_call(blk, x, skips, temb, ehs):
    with torch.no_grad():
        return blk(x.half(), tuple(s.half() for s in skips), temb.half(), encoder_hidden_states=ehs.half()).float()


def test_native_seine_decoder_hooks_vs_reference_fixture(monkeypatch):
    emu.install(monkeypatch)
    sn, blocks = _native_blocks()
    fx = torch.load(FIXTURE)
    out = spec.run_cases(blocks, sn, _native_call)
    for i in spec.BLOCKS:
        assert torch.equal(out[f"block{i}_nohook"], out[f"block{i}_hook_t101"])
        for case in ["nohook"] + [f"hook_t{t}" for t in spec.TS_CASES]:
            got, ref = out[f"block{i}_{case}"], fx[f"block{i}_{case}"]
            assert got.shape == ref.shape
            err = float((got - ref).abs().max() / ref.abs().max())
            l2 = float((got - ref).norm() / ref.norm())
            assert err < 4e-3 and l2 < 2.5e-3, (i, case, err, l2)
        a = out[f"block{i}_nohook"]
        assert torch.equal(a[:1], out[f"block{i}_hook_t981"][:1])
        # every hook kind changes the result: temporal only / + cross-attention / + spatial (+ conv in block 1)
        assert float((out[f"block{i}_hook_t301"][1:] - a[1:]).abs().max() / a.abs().max()) > 0.05
        assert float((out[f"block{i}_hook_t501"][1:] - out[f"block{i}_hook_t301"][1:]).abs().max() / a.abs().max()) > 0.05
        assert float((out[f"block{i}_hook_t981"][1:] - out[f"block{i}_hook_t501"][1:]).abs().max() / a.abs().max()) > 0.05


def test_seine_hook_registration_targets():
    """``seine/pnp_utils.py:121-147,195-196,282-294,363-376,448-458``: the schedule lands on attn1 / attn2 / attn_temp of decoder blocks
    4-11, every other transformer block is switched off (empty schedule), ``t`` is written to all of them and to the conv site."""
    sn, blocks = _native_blocks()
    model = types.SimpleNamespace(unet=spec.StubUNet(blocks))
    sn.register_conv_injection(model, [981])
    sn.register_spatial_attention_pnp(model, torch.tensor([981, 961]))
    sn.register_cross_attention_pnp(model, [981])
    sn.register_temp_attention_pnp(model, [981])
    sn.register_time(model, torch.tensor(981))
    up = model.unet.up_blocks
    assert up[1].resnets[1].injection_schedule == frozenset([981]) and up[1].resnets[1].t == 981
    assert up[1].resnets[0].injection_schedule is None
    for res in (1, 2, 3):
        for b in (0, 1, 2):
            blk = up[res].attentions[b].transformer_blocks[0]
            for name in ("attn1", "attn2", "attn_temp"):
                m = getattr(blk, name)
                assert m.t == 981
                assert m.injection_schedule == (frozenset() if (res == 1 and b == 0) else
                                                frozenset([981, 961]) if name == "attn1" else frozenset([981]))
    assert model.unet.down_blocks[2].attentions[1].transformer_blocks[0].attn_temp.t == 981
    assert model.unet.mid_block.attentions[0].transformer_blocks[0].attn2.t == 981


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_seine_fixture_is_what_the_reference_code_produces_and_keys_match():
    warnings.filterwarnings("ignore")
    att, ublocks, res, pnp, Rotary = ref_stubs.load_reference_seine_decoder()
    rot = Rotary(32)
    ref_blocks = {i: spec.fill_weights(ublocks.CrossAttnUpBlock3D(rotary_emb=rot, **spec.block_kwargs(i)), spec.WEIGHT_SEED).eval()
                  for i in spec.BLOCKS}
    sn, nat_blocks = _native_blocks()
    for i in spec.BLOCKS:
        rs, ns = ref_blocks[i].state_dict(), nat_blocks[i].state_dict()
        assert sorted(rs.keys()) == sorted(ns.keys())
        assert all(tuple(rs[k].shape) == tuple(ns[k].shape) for k in rs)

    def call(blk, x, skips, temb, ehs):
        with torch.no_grad():
            return blk(x, skips, temb, encoder_hidden_states=ehs, use_image_num=0)
    out = spec.run_cases(ref_blocks, pnp, call)
    fx = torch.load(FIXTURE)
    for k, v in fx.items():
        if k != "spec":
            assert torch.allclose(out[k], v, rtol=1e-5, atol=1e-5 * float(v.abs().max())), k
    # the native bias table == the reference's RelativePositionBias.forward
    ta = ref_blocks[1].attentions[0].transformer_blocks[0].attn_temp
    nt = nat_blocks[1].attentions[0].transformer_blocks[0].attn_temp
    for n in (4, 16, 40):
        assert torch.allclose(ta.time_rel_pos_bias(n, device="cpu").float(), nt.time_rel_pos_bias.table(n, "cpu"), atol=2e-3)


# ------------------------------------------------------------------------------------------------- the whole UNet
UNET_FIXTURE = os.path.join(HERE, "golden", "seine_unet.pt")


def _native_unet():
    from anyv2v_amd import seine as sn
    return sn, spec.fill_weights(sn.UNet3DConditionModel(**spec.UNET_CFG), spec.WEIGHT_SEED)


def _native_unet_call(u, sample, t, ehs):
    return u(sample.half(), t, encoder_hidden_states=ehs.half()).sample.float()


def test_native_seine_unet_vs_reference_fixture(monkeypatch):
    """``UNet3DConditionModel.forward`` (``seine/models/unet.py:365-513``) un-hooked and under the four hook families."""
    emu.install(monkeypatch)
    sn, unet = _native_unet()
    fx = torch.load(UNET_FIXTURE)
    out = spec.run_unet_cases(unet, sn, _native_unet_call)
    assert torch.equal(out["unet_nohook_t101"], out["unet_hook_t101"])
    for case in ["nohook"] + [f"hook_t{t}" for t in spec.TS_CASES]:
        got, ref = out[f"unet_{case}"], fx[f"unet_{case}"]
        assert got.shape == ref.shape == (spec.B, 4, spec.UNET_F, spec.UNET_H, spec.UNET_W)
        err = float((got - ref).abs().max() / ref.abs().max())
        l2 = float((got - ref).norm() / ref.norm())
        assert err < 8e-3 and l2 < 4e-3, (case, err, l2)
    a = out["unet_nohook"]
    assert torch.equal(a[:1], out["unet_hook_t981"][:1])
    assert float((a[1:] - out["unet_hook_t981"][1:]).abs().max() / a.abs().max()) > 0.05


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_seine_unet_fixture_is_what_the_reference_code_produces_and_keys_match():
    warnings.filterwarnings("ignore")
    att, ublocks, res, pnp, Rotary = ref_stubs.load_reference_seine_decoder(with_unet=True)
    ref = spec.fill_weights(ublocks.unet.UNet3DConditionModel(**spec.UNET_CFG), spec.WEIGHT_SEED).eval()
    sn, nat = _native_unet()
    rs, ns = ref.state_dict(), nat.state_dict()
    assert sorted(rs.keys()) == sorted(ns.keys())
    assert all(tuple(rs[k].shape) == tuple(ns[k].shape) for k in rs)
    nat.load_state_dict(rs, strict=True)

    def call(u, sample, t, ehs):
        with torch.no_grad():
            return u(sample, t, encoder_hidden_states=ehs).sample
    out = spec.run_unet_cases(ref, pnp, call)
    fx = torch.load(UNET_FIXTURE)
    for k, v in fx.items():
        if k != "spec":
            assert torch.allclose(out[k], v, rtol=1e-5, atol=1e-5 * float(v.abs().max())), k


# ------------------------------------------------------------------------------------------------- the two runner classes
PIPE_FIXTURE = os.path.join(HERE, "golden", "seine_pipeline.pt")
# cfg_scale 4 multiplies the difference of two branch predictions (and its fp16 rounding) by 4, over 4 steps on latents that random
# weights drive to |x| ~ 10-17: measured on the op emulation 2.2e-2 (DDIM) / 1.2e-2 (DDPM) of the range
EDIT_TOL = 6e-2


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max())


def _check_job(nat, ref, sm):
    assert _rel(nat["lat0"], ref["lat0"]) <= 2e-3
    for i, t in enumerate(ref["inv_ts"]):
        e = _rel(nat["files"][t], ref["trajectory"][i])
        assert e <= 1e-2, (f"inversion file t={t}", e)
    assert sorted(nat["files"]) == ref["inv_ts"]
    assert _rel(nat["recon_lat"], ref["recon_lat"]) <= 2e-2
    assert nat["edit_ts"] == ref[f"edit_ts_{sm}"]
    e = _rel(nat["edit_lat"], ref[f"edit_lat_{sm}"])
    assert e <= EDIT_TOL, (f"edit ({sm})", e)
    # decode_latents on the REFERENCE's latents: scaling, frame order, uint8 conversion ((x / 2 + 0.5) * 255 + 0.5, truncated)
    dec = nat["pipe"].decode_latents(ref[f"edit_lat_{sm}"].to(nat["edit_lat"].device))
    assert dec.shape == ref[f"edited_frames_{sm}"].shape and dec.dtype == torch.uint8
    assert int((dec.int() - ref[f"edited_frames_{sm}"].int()).abs().max()) <= 1


@pytest.mark.parametrize("sm", ["ddim", "ddpm"])
def test_native_seine_runner_classes_vs_reference_fixture(monkeypatch, tmp_path, sm):
    """``SEINEDDIMInversionPipeline`` / ``SEINEPnPPipeline`` on the op emulation vs the fixture the reference's own runner classes produced
    (``make_golden.py --seine-pipeline``): frame pre-processing + VAE latents, every trajectory file, the DDIM reconstruction, the PnP edit
    with the DDIM sampler and with the shipped default (ancestral DDPM, same noise stream), decoding."""
    emu.install(monkeypatch)
    fx = torch.load(PIPE_FIXTURE)
    files = {t: fx["trajectory"][i] for i, t in enumerate(fx["inv_ts"])}
    job = spec.native_job("cpu", tmp_path, sm, trajectory_from=files)
    _check_job(job, fx, sm)
    if sm == "ddim":
        # ``compute_masked_video_latents_at_0(config, video_input)`` with the reference's [b, f, c, H, W] argument (``run_pnp_edit.py:256``)
        # and with the first frame alone: the same mask and latents (a toy VAE without posterior noise)
        p2 = job["pipe"]
        first = p2.src_video_frames[0].unsqueeze(0)
        clip = torch.cat([first, torch.randn(len(p2.src_video_frames) - 1, *first.shape[1:])]).unsqueeze(0)     # (later frames are masked out)
        m5, l5 = p2.compute_masked_video_latents_at_0(p2.config, clip)
        m4, l4 = p2.compute_masked_video_latents_at_0(p2.config, first)
        assert torch.equal(m5, m4) and torch.equal(l5, l4)
        with pytest.raises(NotImplementedError):
            p2.compute_masked_video_latents_at_0(p2.config, torch.cat([clip, clip]))


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_seine_pipeline_fixture_is_what_the_reference_runner_classes_produce(tmp_path):
    warnings.filterwarnings("ignore")
    from oracle import ref_seine_pipeline as rsp
    frames, edited = spec.job_frames()
    fx = torch.load(PIPE_FIXTURE)
    for sm in ("ddim", "ddpm"):
        inv, ed = spec.job_configs(sm)
        job = rsp.run_reference_job(spec.UNET_CFG, spec.fill_weights, spec.WEIGHT_SEED, frames, edited, inv, ed, tmp_path / sm)
        assert _rel(fx[f"edit_lat_{sm}"], job["edit_lat"]) <= 2e-3 and torch.equal(fx[f"edited_frames_{sm}"], job["edited_frames"])
        assert job["edit_ts"] == fx[f"edit_ts_{sm}"]
        if sm == "ddim":
            assert _rel(fx["lat0"], job["lat0"]) <= 1e-3 and _rel(fx["recon_lat"], job["recon_lat"]) <= 2e-3
            for i, t in enumerate(job["inv_ts"]):
                assert _rel(fx["trajectory"][i], job["files"][t]) <= 2e-3


# ------------------------------------------------------------------------------------------------- the CLI runners
def run_seine_cli_stages(base, device):
    """Both CLIs as two processes would run them, on a toy checkpoint directory (SD-style ``unet/config.json`` + ``seine.pt``)."""
    import json
    from anyv2v_amd import seine as sn
    from anyv2v_amd import seine_run_ddim_inversion as s1, seine_run_pnp_edit as s2
    base = str(base)
    j = spec.JOB
    os.makedirs(os.path.join(base, "sd", "unet"), exist_ok=True)
    json.dump(dict(spec.UNET_CFG, _class_name="UNet2DConditionModel"), open(os.path.join(base, "sd", "unet", "config.json"), "w"))
    unet = spec.fill_weights(sn.UNet3DConditionModel(**spec.UNET_CFG), spec.WEIGHT_SEED)
    torch.save({"ema": unet.state_dict()}, os.path.join(base, "seine.pt"))
    frames, edited = spec.job_frames()
    os.makedirs(os.path.join(base, "clip"), exist_ok=True)
    for i, f in enumerate(frames):
        f.save(os.path.join(base, "clip", f"{i:05d}.png"))
    edited.save(os.path.join(base, "edited.png"))
    common = [f"device={device}", f"sd_path={base}/sd", f"ckpt_path={base}/seine.pt", f"image_size=[{j['height']},{j['width']}]"]
    root = os.path.join(os.path.dirname(HERE), "configs", "seine")
    save1 = s1.cli(["--config", os.path.join(root, "ddim_inversion.yaml"), "--video_path", os.path.join(base, "clip")] + common +
                   [f"output_dir={base}/ddim-inversion/default", f"n_steps={j['inv_steps']}", f"n_save_steps={j['save_steps']}",
                    f"n_frame_to_invert={j['frames']}"])
    save2 = s2.cli(["--config", os.path.join(root, "pnp_edit.yaml")] + common +
                   [f"output_dir={base}/results", f"src_video_path={base}/clip.mp4", f"edited_first_frame_path={base}/edited.png",
                    f"ddim_inversion_dir={base}/ddim-inversion/default", f"n_ddim_inversion_steps={j['inv_steps']}", f"n_frame_inverted={j['frames']}",
                    f"n_frames={j['frames']}", f"n_steps={j['edit_steps']}", "prompt=a robot", "pnp_f_t=0.5", "pnp_spatial_attn_t=0.5",
                    "pnp_cross_attn_t=0.25", "pnp_temp_attn_t=0.75"])
    return save1, save2


def check_seine_cli_outputs(save1, save2):
    import numpy as np
    from PIL import Image
    from anyv2v_amd.mp4 import read_mp4
    j = spec.JOB
    assert save1.endswith(os.path.join("seine", "clip", f"steps_{j['inv_steps']}", f"nframes_{j['frames']}"))
    lat = sorted(os.listdir(os.path.join(save1, "ddim_latents")))
    assert lat == sorted(f"ddim_latents_{t}.pt" for t in (1, 251, 501, 751))
    x = torch.load(os.path.join(save1, "ddim_latents", "ddim_latents_751.pt"))
    assert tuple(x.shape) == (1, 4, j["frames"], j["height"] // 8, j["width"] // 8) and torch.isfinite(x.float()).all()
    import yaml
    assert yaml.safe_load(open(os.path.join(save1, "inversion_prompts.yaml"))) == {"clip": ""}
    assert yaml.safe_load(open(os.path.join(save1, "config.yaml")))["n_steps"] == j["inv_steps"]
    assert len(os.listdir(os.path.join(save1, "recon_frames"))) == j["frames"] and len(read_mp4(os.path.join(save1, "inverted.mp4"))[0]) == j["frames"]
    assert save2.endswith(os.path.join("seine", "clip", "a_robot", f"cfg4_f0.5_spa0.5_cro0.25_tmp0.75_stp{j['edit_steps']}"))
    pngs = sorted(os.listdir(os.path.join(save2, "img_ode")))
    assert pngs == [f"{i:05d}.png" for i in range(j["frames"])]
    vid, fps = read_mp4(os.path.join(save2, "video_pnp_fps_8.mp4"))
    assert len(vid) == j["frames"] and vid[0].size == (j["width"], j["height"]) and fps == 8.0
    return np.stack([np.asarray(Image.open(os.path.join(save2, "img_ode", p))) for p in pngs])


def test_seine_cli_runners_end_to_end(monkeypatch, tmp_path):
    """``seine/run_ddim_inversion.py`` -> ``run_pnp_edit.py`` as CLIs: the reference's config files and directory layout (the edit finds
    the inversion by globbing ``<ddim_inversion_dir>/seine/<clip>/steps_<n>/nframes_*``), a checkpoint in the reference's format
    (``{"ema": state_dict}``), PNG frames in; latents, yaml files, frames and mp4 out; same seed -> same frames (DDPM sampler)."""
    emu.install(monkeypatch)
    a = check_seine_cli_outputs(*run_seine_cli_stages(tmp_path / "a", "cpu"))
    b = check_seine_cli_outputs(*run_seine_cli_stages(tmp_path / "b", "cpu"))
    assert (a == b).all()


@pytest.mark.gpu
def test_seine_cli_runners_on_gpu(tmp_path):
    check_seine_cli_outputs(*run_seine_cli_stages(tmp_path / "a", "cuda:0"))


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_ddpm_step_is_pinned_to_the_references_own_gaussian_diffusion():
    """diffusers' ``DDPMScheduler`` is not in the reference tree, but SEINE vendors the process it implements:
    ``seine/diffusion/gaussian_diffusion.py`` (``q_posterior_mean_variance`` ``:235-255``, ``p_mean_variance`` ``:257-390``, FIXED_SMALL
    variance) with ``respace.SpacedDiffusion`` (``:66-93``: the betas of a sub-sequence of timesteps).  On the 50 timesteps the edit visits
    its posterior mean and variance ARE the ancestral step: ``schedulers.DDPMScheduler``'s coefficients are checked against them."""
    import importlib.util
    import sys
    from anyv2v_amd.schedulers import SEINE_SCHEDULER_CONFIG, DDPMScheduler
    root = os.path.join(ref_stubs.REFERENCE_ROOT, "seine", "diffusion")
    spec_ = importlib.util.spec_from_file_location("_ref_seine_diffusion", os.path.join(root, "__init__.py"), submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec_)
    sys.modules["_ref_seine_diffusion"] = mod
    try:
        spec_.loader.exec_module(mod)
        gd = sys.modules["_ref_seine_diffusion.gaussian_diffusion"]
        sched = DDPMScheduler(**SEINE_SCHEDULER_CONFIG)
        sched.set_timesteps(50)
        ts = sorted(int(t) for t in sched.timesteps)
        assert ts == list(range(0, 1000, 20))
        diff = mod.SpacedDiffusion(use_timesteps=ts, betas=gd.get_named_beta_schedule("linear", 1000), model_mean_type=gd.ModelMeanType.EPSILON,
                                   model_var_type=gd.ModelVarType.FIXED_SMALL, loss_type=gd.LossType.MSE)
        g = torch.Generator().manual_seed(0)
        x, eps = torch.randn(2, 4, 3, 5, 5, generator=g, dtype=torch.float64), torch.randn(2, 4, 3, 5, 5, generator=g, dtype=torch.float64)
        for i, t in enumerate(ts):
            if i not in (0, 1, 7, 25, 49):
                continue
            out = diff.p_mean_variance(lambda xx, tt: eps, x, torch.tensor([i, i]), clip_denoised=False)
            sa_t, sb_t, cx, ce, sigma = sched.ancestral_coefficients(t)
            x0 = (x - sb_t * eps) / sa_t
            mean = cx * x0 + ce * eps
            assert torch.allclose(mean, out["mean"], rtol=1e-4, atol=1e-5), t      # (fp32 alpha table here, as in diffusers; fp64 numpy there)
            assert torch.allclose(x0, out["pred_xstart"], rtol=1e-4, atol=1e-5), t
            if t > 0:
                assert abs(sigma ** 2 - float(out["variance"].flatten()[0])) <= 5e-4 * sigma ** 2, t   # (1 - a_prev of an fp32 table next to 1: 1.6e-4 at t = 20)
            else:
                assert sigma == 0.0     # (p_sample masks the noise at t = 0, gaussian_diffusion.py:431-433)
    finally:
        for k in [k for k in sys.modules if k.startswith("_ref_seine_diffusion")]:
            del sys.modules[k]


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
@pytest.mark.parametrize("hw", [(9, 10), (12, 12), (13, 9)])
def test_native_seine_unet_at_latent_sizes_that_are_not_multiples_of_8_vs_the_references_class(monkeypatch, hw):
    """``seine/models/unet.py:393-401,485-500`` + ``resnet.py:44-64``: a latent size that three ceil-halvings do not give back by doubling
    makes the UNet hand every up block the size of the skip connections ahead (nearest-neighbour resize to exactly that size).  The
    reference's own class vs the native UNet; the same rule serves the I2VGen-XL and ConsistI2V UNets (``unet.upsample_tokens``)."""
    warnings.filterwarnings("ignore")
    from anyv2v_amd import seine as sn
    att, ublocks, res, pnp, Rotary = ref_stubs.load_reference_seine_decoder(with_unet=True)
    ref = spec.fill_weights(ublocks.unet.UNet3DConditionModel(**spec.UNET_CFG), spec.WEIGHT_SEED).eval()
    emu.install(monkeypatch)
    nat = spec.fill_weights(sn.UNet3DConditionModel(**spec.UNET_CFG), spec.WEIGHT_SEED)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 9, spec.UNET_F, *hw, generator=g).half().float()
    ehs = torch.randn(2, spec.TOKENS, spec.CROSS, generator=g).half().float()
    with torch.no_grad():
        want = ref(x, 981, encoder_hidden_states=ehs).sample
        got = nat(x.half(), 981, encoder_hidden_states=ehs.half()).sample
    assert got.shape == want.shape
    err = float((got.float() - want).abs().max() / want.abs().max())
    assert err < 8e-3, err


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_shipped_configs_resolve_to_the_references_values():
    """``configs/seine/*.yaml``: another layout than the reference's files, the same keys and values."""
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("ddim_inversion.yaml", "pnp_edit.yaml"):
        mine = yaml.safe_load(open(os.path.join(root, "configs", "seine", name)))
        ref = yaml.safe_load(open(os.path.join(ref_stubs.REFERENCE_ROOT, "seine", "configs", name)))
        assert mine == ref, name


def test_seine_helper_functions_of_the_references_pnp_utils(tmp_path):
    """``save_video_as_frames`` / ``load_imgs`` / ``save_video`` / ``load_video_frames`` (``seine/pnp_utils.py:25-118``) under their own names."""
    import numpy as np
    from PIL import Image
    from anyv2v_amd import seine_pipeline as sp
    from anyv2v_amd.mp4 import read_mp4
    yy, xx = np.mgrid[0:16, 0:24]
    frames = torch.from_numpy(np.stack([np.stack([(xx * 8 + 20 * i) % 200, yy * 12, (xx + yy) * 5 + 10 * i]) for i in range(3)]).astype(np.uint8))
    path = str(tmp_path / "clip.mp4")
    sp.save_video(frames, path, fps=8)
    vid, fps = read_mp4(path)
    close = lambda img, t: np.abs(np.asarray(img).astype(int) - t.permute(1, 2, 0).numpy().astype(int)).mean() < 6      # (4:2:0 chroma)
    assert fps == 8.0 and len(vid) == 3 and vid[0].size == (24, 16) and close(vid[1], frames[1])
    sp.save_video(frames.float() / 255.0 + 1e-4, str(tmp_path / "clip01.mp4"), fps=10, scaling_255=True)
    vid01, fps01 = read_mp4(str(tmp_path / "clip01.mp4"))
    assert fps01 == 10.0 and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(vid, vid01))
    sp.save_video_as_frames(path, img_size=(12, 8))
    out = str(tmp_path / "clip")
    assert sorted(os.listdir(out)) == ["00000.png", "00001.png", "00002.png"] and Image.open(os.path.join(out, "00000.png")).size == (12, 8)
    imgs, pils = sp.load_imgs(out, 3, device="cpu", pil=True)
    assert tuple(imgs.shape) == (3, 3, 8, 12) and imgs.dtype == torch.float32 and 0 <= float(imgs.min()) and float(imgs.max()) <= 1 and len(pils) == 3
    paths, u8 = sp.load_video_frames(out, 3)
    assert u8.dtype == torch.uint8 and tuple(u8.shape) == (3, 3, 8, 12) and torch.equal((imgs * 255).round().to(torch.uint8), u8)
