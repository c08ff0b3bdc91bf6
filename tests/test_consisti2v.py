"""SURVEY.md 8(f) F4 -- the ConsistI2V hook family (``anyv2v_amd/consisti2v.py``) on the CPU: native decoder blocks through the op
emulation vs the fixture produced by the REFERENCE's own ``VideoLDMCrossAttnUpBlock`` + ``consisti2v/pnp_utils.py``
(``tests/golden/make_golden.py --consisti2v``), and -- where /root/reference exists -- the fixture vs the reference run live."""
import os
import warnings

import pytest
import torch

import consisti2v_spec as spec
import cpu_ops_emulation as emu
from oracle import ref_stubs

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "consisti2v_decoder_hooks.pt")


def _native_blocks():
    from anyv2v_amd import consisti2v as c2
    return c2, {i: spec.fill_weights(c2.VideoLDMCrossAttnUpBlock(**spec.block_kwargs(i))) for i in spec.BLOCKS}


def _native_call(blk, x, skips, temb, ehs):
    with torch.no_grad():
        return blk(x.half(), tuple(s.half() for s in skips), temb.half(), encoder_hidden_states=ehs.half()).float()


def test_native_consisti2v_decoder_hooks_vs_reference_fixture(monkeypatch):
    emu.install(monkeypatch)
    c2, blocks = _native_blocks()
    fx = torch.load(FIXTURE)
    out = spec.run_cases(blocks, c2, _native_call)
    for i in spec.BLOCKS:
        # a timestep outside every schedule leaves the blocks un-hooked (bit-equal on the native path as well)
        assert torch.equal(out[f"block{i}_nohook"], out[f"block{i}_hook_t101"])
        for case in ["nohook"] + [f"hook_t{t}" for t in spec.TS_CASES]:
            got, ref = out[f"block{i}_{case}"], fx[f"block{i}_{case}"]
            assert got.shape == ref.shape
            err = float((got - ref).abs().max() / ref.abs().max())
            l2 = float((got - ref).norm() / ref.norm())
            assert err < 4e-3 and l2 < 2e-3, (i, case, err, l2)   # fp16 activations / weights vs the fp32 reference
        # the hooks do something, and only to the two injected branches
        third = spec.B * spec.FR // 3
        a, h = out[f"block{i}_nohook"], out[f"block{i}_hook_t981"]
        assert torch.equal(a[:third], h[:third])
        assert float((a[third:] - h[third:]).abs().max() / a.abs().max()) > 0.1


def test_hook_registration_targets_match_the_reference_paths():
    """``consisti2v/pnp_utils.py:20-28,130,228-240,356-362``: which modules get a schedule / a processor / a timestep."""
    c2, blocks = _native_blocks()
    model = spec.stub_model(blocks)
    c2.register_conv_injection(model, [981])
    c2.register_spatial_attention_pnp(model, torch.tensor([981, 961]))
    c2.register_temp_attention_pnp(model, [981])
    c2.register_time(model, torch.tensor(981))
    up = model.unet.up_blocks
    assert up[1].resnets[1].injection_schedule == frozenset([981]) and up[1].resnets[1].t == 981
    assert up[1].resnets[0].injection_schedule is None and up[2].resnets[1].injection_schedule is None
    for res in (1, 2, 3):
        for blk in (0, 1, 2):
            injected = not (res == 1 and blk == 0)    # decoder blocks 4-11: not the first block of the lowest resolution
            for tree, cls in ((up[res].attentions, c2.HipSpaAttnProcessor), (up[res].tempo_attns, c2.HipTmpAttnProcessor)):
                proc = tree[blk].transformer_blocks[0].attn1.processor
                assert isinstance(proc, cls) and proc.t == 981
                assert (proc.injection_schedule is not None) == injected
            assert up[res].attentions[blk].transformer_blocks[0].attn2.processor.injection_schedule is None


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_fixture_is_what_the_reference_code_produces_and_keys_match():
    warnings.filterwarnings("ignore")
    att, blocks_mod, ublocks, pnp = ref_stubs.load_reference_consisti2v_decoder()
    ref_blocks = {i: spec.fill_weights(ublocks.VideoLDMCrossAttnUpBlock(**spec.block_kwargs(i))).eval() for i in spec.BLOCKS}
    c2, nat_blocks = _native_blocks()
    for i in spec.BLOCKS:   # same module tree: state-dict keys and shapes
        rs, ns = ref_blocks[i].state_dict(), nat_blocks[i].state_dict()
        assert sorted(rs.keys()) == sorted(ns.keys())
        assert all(tuple(rs[k].shape) == tuple(ns[k].shape) for k in rs)

    def call(blk, x, skips, temb, ehs):
        with torch.no_grad():
            return blk(x, skips, temb, encoder_hidden_states=ehs)
    out = spec.run_cases(ref_blocks, pnp, call)
    fx = torch.load(FIXTURE)
    for k, v in fx.items():
        if k != "spec":
            assert torch.allclose(out[k], v, rtol=1e-5, atol=1e-5 * float(v.abs().max())), k


# ------------------------------------------------------------------------------------------------- the whole UNet
UNET_FIXTURE = os.path.join(HERE, "golden", "consisti2v_unet.pt")


def _native_unet():
    from anyv2v_amd import consisti2v as c2
    return c2, spec.fill_weights(c2.VideoLDMUNet3DConditionModel(**spec.UNET_CFG))


def _native_unet_call(u, sample, t, ehs, first, stride):
    return u(sample.half(), t, encoder_hidden_states=ehs.half(), first_frame_latents=first.half(), frame_stride=stride).sample.float()


def test_native_consisti2v_unet_vs_reference_fixture(monkeypatch):
    """``VideoLDMUNet3DConditionModel.forward`` (``videoldm_unet.py:687-1026``) un-hooked and under the PnP hooks."""
    emu.install(monkeypatch)
    c2, unet = _native_unet()
    fx = torch.load(UNET_FIXTURE)
    out = spec.run_unet_cases(unet, c2, _native_unet_call)
    assert torch.equal(out["unet_nohook_t101"], out["unet_hook_t101"])
    for case in ["nohook"] + [f"hook_t{t}" for t in spec.TS_CASES]:
        got, ref = out[f"unet_{case}"], fx[f"unet_{case}"]
        assert got.shape == ref.shape == (spec.B, 4, spec.UNET_CFG["n_frames"] - 1, spec.UNET_H, spec.UNET_W)
        err = float((got - ref).abs().max() / ref.abs().max())
        l2 = float((got - ref).norm() / ref.norm())
        assert err < 8e-3 and l2 < 4e-3, (case, err, l2)   # fp16 activations / weights through ~60 layers vs the fp32 reference
    a, h = out["unet_nohook"], out["unet_hook_t981"]
    assert torch.equal(a[:1], h[:1])                         # the source branch never sees the hooks
    assert float((a[1:] - h[1:]).abs().max() / a.abs().max()) > 0.05


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_unet_fixture_is_what_the_reference_code_produces_and_keys_match():
    warnings.filterwarnings("ignore")
    unet_mod, ublocks, pnp = ref_stubs.load_reference_consisti2v_unet()
    ref = spec.fill_weights(unet_mod.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)).eval()
    c2, nat = _native_unet()
    rs, ns = ref.state_dict(), nat.state_dict()
    assert sorted(rs.keys()) == sorted(ns.keys())
    assert all(tuple(rs[k].shape) == tuple(ns[k].shape) for k in rs)
    nat.load_state_dict(rs, strict=True)                     # a reference checkpoint loads unchanged

    def call(u, sample, t, ehs, first, stride):
        with torch.no_grad():
            return u(sample, t, encoder_hidden_states=ehs, first_frame_latents=first, frame_stride=stride).sample
    out = spec.run_unet_cases(ref, pnp, call)
    fx = torch.load(UNET_FIXTURE)
    for k, v in fx.items():
        if k != "spec":
            assert torch.allclose(out[k], v, rtol=1e-5, atol=1e-5 * float(v.abs().max())), k


# ------------------------------------------------------------------------------------------------- the pipeline
PIPE_FIXTURE = os.path.join(HERE, "golden", "consisti2v_pipeline.pt")


# Text guidance 35 (the reference's configs/pipeline_256/pnp_edit.yaml:23) multiplies the difference of two branch predictions -- and
# their fp16 rounding -- by 35: measured on the op emulation the edit's error is 0.007 / 0.045 / 0.15 of the latents' range at
# guidance 2 / 9 / 35, i.e. linear in the scale.  The per-branch error itself is bounded (un-amplified) by the UNet fixture above.
EDIT_TOL = 0.25


def _close(a, b, tol):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max()) <= tol * float(b.abs().max()), float((a - b).abs().max() / b.abs().max())


def _check_decode(nat, ref_lat, ref_video):
    """``decode_latents`` (scaling, frame-by-frame decode, [0, 1] range) on the REFERENCE's edited latents vs the reference's video; and
    ``output_type="tensor"`` is that decode of the native latents."""
    dev = nat["edit_lat"].device
    ok, err = _close(torch.from_numpy(nat["pipe"].decode_latents(ref_lat.to(dev).half())), ref_video, 4e-3)
    assert ok, ("decode_latents", err)
    assert torch.equal(nat["edit_video"], torch.from_numpy(nat["pipe"].decode_latents(nat["edit_lat"])))


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_native_consisti2v_pipeline_vs_the_references_own_pipeline_class(monkeypatch, tmp_path):
    """``ConditionalVideoEditingPipeline`` end to end: the REFERENCE's class (verbatim, its own UNet / hooks / inverse scheduler /
    ``load_ddim_latents_at_t``) vs the native pipeline (op emulation) from the same PNG frames and prompt strings with the same toy
    VAE / text-encoder weights: ``encode_vae_video``, the inversion trajectory, the reconstruction, the PnP edit.  The committed
    fixture (what the -m gpu suite compares the HIP path with) must be what the reference produces."""
    warnings.filterwarnings("ignore")
    from oracle import ref_consisti2v_pipeline as rcp
    j = spec.PIPE_JOB
    frames, edited = spec.pipeline_frames()
    job = rcp.run_reference_job(spec.UNET_CFG, spec.fill_weights, frames, edited, j["height"], j["width"], j["n_inv_steps"], j["n_steps"],
                                j["t_idx"], j["ratios"], tmp_path, frame_stride=j["frame_stride"], edit_prompt=j["edit_prompt"],
                                neg=j["neg"], cfg_txt=j["cfg_txt"])
    fx = torch.load(PIPE_FIXTURE)
    for k in ("lat0", "rec_lat", "edit_lat", "edit_video"):
        ok, err = _close(fx[k], job[k], 2e-3)          # fixture tensors are stored in fp16
        assert ok, (k, err)
    for i, t in enumerate(job["inv_ts"]):
        ok, err = _close(fx["trajectory"][i], job["files"][t], 2e-3)
        assert ok, (t, err)
    emu.install(monkeypatch)
    # stage by stage: every native stage starts from the reference's own trajectory, so that errors do not compound across stages
    nat = spec.native_pipeline_job("cpu", trajectory_from=job["files"])
    ok, err = _close(nat["lat0"], job["lat0"], 2e-3)
    assert ok, ("encode_vae_video (Resize + CenterCrop + per-frame encode)", err)
    assert nat["inv_ts"] == job["inv_ts"]
    for t in job["inv_ts"]:
        ok, err = _close(nat["files"][t], job["files"][t], 2e-2)
        assert ok, (f"invert at t={t}", err)
    for k, tol in (("rec_lat", 2e-2), ("edit_lat", EDIT_TOL)):
        ok, err = _close(nat[k], job[k], tol)
        assert ok, (k, err)
    _check_decode(nat, job["edit_lat"], job["edit_video"])
    # the hooks matter: the edit is far from the reconstruction
    assert float((job["edit_lat"] - job["rec_lat"]).abs().max()) > 0.5


def test_native_consisti2v_pipeline_vs_reference_fixture(monkeypatch):
    """The same comparison against the committed fixture (runs where /root/reference does not exist)."""
    emu.install(monkeypatch)
    fx = torch.load(PIPE_FIXTURE)
    files = {t: fx["trajectory"][i] for i, t in enumerate(fx["inv_ts"])}
    nat = spec.native_pipeline_job("cpu", trajectory_from=files)
    assert nat["inv_ts"] == fx["inv_ts"]
    for i, t in enumerate(fx["inv_ts"]):
        ok, err = _close(nat["files"][t], fx["trajectory"][i], 2e-2)
        assert ok, (f"invert at t={t}", err)
    for k, tol in (("lat0", 3e-3), ("rec_lat", 2e-2), ("edit_lat", EDIT_TOL)):
        ok, err = _close(nat[k], fx[k], tol)
        assert ok, (k, err)
    _check_decode(nat, fx["edit_lat"], fx["edit_video"])


# ------------------------------------------------------------------------------------------------- the CLI runners
def _consisti2v_workspace(base):
    """<base>/model (toy checkpoint: unet/config.json + safetensors in the reference's key naming), <base>/clip (PNG frames)."""
    import json

    import numpy as np
    from PIL import Image
    from safetensors.torch import save_file
    from anyv2v_amd import consisti2v as c2
    base = str(base)
    os.makedirs(os.path.join(base, "model", "unet"), exist_ok=True)
    unet = spec.fill_weights(c2.VideoLDMUNet3DConditionModel(**spec.UNET_CFG))
    save_file({k: v.contiguous() for k, v in unet.state_dict().items()}, os.path.join(base, "model", "unet", "diffusion_pytorch_model.safetensors"))
    json.dump(dict(spec.UNET_CFG, _class_name="VideoLDMUNet3DConditionModel"), open(os.path.join(base, "model", "unet", "config.json"), "w"))
    frames, edited = spec.pipeline_frames()
    os.makedirs(os.path.join(base, "clip"), exist_ok=True)
    W, H = spec.PIPE_JOB["width"], spec.PIPE_JOB["height"]
    for i, f in enumerate(frames):
        f.resize((W, H), resample=Image.Resampling.LANCZOS).save(os.path.join(base, "clip", f"{i:05d}.png"))
    edited.save(os.path.join(base, "edited.png"))
    return base


def run_consisti2v_cli_stages(base, device):
    """Stage 1 and stage 2 through the two CLIs, as two processes would run them (``python -m anyv2v_amd.consisti2v_run_*``)."""
    from anyv2v_amd import consisti2v_run_ddim_inversion as s1, consisti2v_run_pnp_edit as s2
    j = spec.PIPE_JOB
    common = [f"device={device}", f"model_path={base}/model", "video_name=clip", f"video_path={base}/missing.mp4",
              f"video_frames_path={base}/clip", f"image_size=[{j['width']},{j['height']}]", f"n_frames={j['frames']}"]
    s1.cli(["--config", os.path.join(ROOT_DIR, "configs", "consisti2v", "pipeline_256", "ddim_inversion_256.yaml")] + common +
           [f"output_dir={base}/ddim_inversion/clip", f"inverse_config.output_dir={base}/outputs/clip", f"inverse_config.n_steps={j['n_inv_steps']}",
            f"recon_config.n_steps={j['n_steps']}", f"recon_config.ddim_init_latents_t_idx={j['t_idx']}"])
    s2.cli(["--config", os.path.join(ROOT_DIR, "configs", "consisti2v", "pipeline_256", "pnp_edit.yaml")] + common +
           [f"output_dir={base}/results/clip", f"edited_first_frame_path={base}/edited.png", f"ddim_latents_path={base}/outputs",
            f"n_steps={j['n_steps']}", f"ddim_init_latents_t_idx={j['t_idx']}", "editing_prompt=a robot", "pnp_f_t=0.5", "pnp_spatial_attn_t=0.5",
            "pnp_temp_attn_t=0.75"])


ROOT_DIR = os.path.dirname(HERE)


def check_consisti2v_cli_outputs(base):
    from anyv2v_amd.mp4 import read_mp4
    from PIL import Image
    j = spec.PIPE_JOB
    lat = sorted(os.listdir(os.path.join(base, "outputs", "clip")))
    assert len(lat) == j["n_inv_steps"] and all(f.startswith("ddim_latents_") and f.endswith(".pt") for f in lat)
    x = torch.load(os.path.join(base, "outputs", "clip", "ddim_latents_501.pt"))
    assert tuple(x.shape) == (1, 4, j["frames"], j["height"] // 8, j["width"] // 8) and torch.isfinite(x.float()).all()
    for d, fps in ((os.path.join(base, "ddim_inversion", "clip", "ddim_reconstruction"), 10.0), (os.path.join(base, "results", "clip", "a robot", "video"), 8.0)):
        vid, f = read_mp4(d + ".mp4")
        assert len(vid) == j["frames"] and vid[0].size == (j["width"], j["height"]) and f == fps
        gif = Image.open(d + ".gif")
        assert gif.n_frames == j["frames"] and gif.size == (j["width"], j["height"])
    rec, _ = read_mp4(os.path.join(base, "ddim_inversion", "clip", "ddim_reconstruction.mp4"))
    ed, _ = read_mp4(os.path.join(base, "results", "clip", "a robot", "video.mp4"))
    import numpy as np
    assert not np.array_equal(np.asarray(rec[1]), np.asarray(ed[1]))
    return np.stack([np.asarray(f) for f in ed])


def test_consisti2v_cli_runners_end_to_end(monkeypatch, tmp_path):
    """``consisti2v/run_ddim_inversion.py`` -> ``run_pnp_edit.py`` as CLIs: config files with the reference's keys and dotlist
    overrides, a checkpoint directory, PNG frames in; ``ddim_latents_{t}.pt``, reconstruction and edited gif / mp4 out; same seed ->
    the same video."""
    emu.install(monkeypatch)
    base = _consisti2v_workspace(tmp_path)
    run_consisti2v_cli_stages(base, "cpu")
    a = check_consisti2v_cli_outputs(base)
    run_consisti2v_cli_stages(base, "cpu")
    b = check_consisti2v_cli_outputs(base)
    assert (a == b).all()


def test_consisti2v_inversion_cli_from_a_video_file(monkeypatch, tmp_path):
    """``run_ddim_inversion.py:106-112``: from ``video_path`` the resized frames are written to ``<output_dir>/<video name>/`` (``save_dir``,
    ``consisti2v/utils.py:55-76``) and the first frame is opened from there; ``save_frames: False`` cannot work with a video file."""
    from PIL import Image
    from anyv2v_amd import consisti2v_run_ddim_inversion as s1
    from anyv2v_amd.utils import export_to_video
    emu.install(monkeypatch)
    base = _consisti2v_workspace(tmp_path)
    j = spec.PIPE_JOB
    frames = [Image.open(os.path.join(base, "clip", f"{i:05d}.png")).convert("RGB") for i in range(j["frames"])]
    big = [f.resize((j["width"] + 32, j["height"] + 16), resample=Image.Resampling.LANCZOS) for f in frames]     # (resized on the way in)
    export_to_video(big, os.path.join(base, "movie.mp4"), fps=8)
    args = ["--config", os.path.join(ROOT_DIR, "configs", "consisti2v", "pipeline_256", "ddim_inversion_256.yaml"), "device=cpu",
            f"model_path={base}/model", "video_name=movie", f"video_path={base}/movie.mp4", f"image_size=[{j['width']},{j['height']}]",
            f"n_frames={j['frames']}", f"output_dir={base}/inv/movie", f"inverse_config.output_dir={base}/outputs/movie",
            f"inverse_config.n_steps={j['n_inv_steps']}", f"recon_config.n_steps={j['n_steps']}", f"recon_config.ddim_init_latents_t_idx={j['t_idx']}"]
    s1.cli(args)
    saved = sorted(os.listdir(os.path.join(base, "inv", "movie", "movie")))
    assert saved == [f"{i:05d}.png" for i in range(j["frames"])]
    assert Image.open(os.path.join(base, "inv", "movie", "movie", "00000.png")).size == (j["width"], j["height"])
    assert not os.path.isdir(os.path.join(base, "movie"))            # (not next to the video: that is stage 2's habit, run_pnp_edit.py:70-74)
    assert len(os.listdir(os.path.join(base, "outputs", "movie"))) == j["n_inv_steps"]
    with pytest.raises(ValueError, match="save_frames"):
        s1.cli(args + ["save_frames=False"])


@pytest.mark.gpu
def test_consisti2v_cli_runners_on_gpu(tmp_path):
    """The same two CLIs on the HIP library."""
    base = _consisti2v_workspace(tmp_path)
    run_consisti2v_cli_stages(base, "cuda:0")
    a = check_consisti2v_cli_outputs(base)
    run_consisti2v_cli_stages(base, "cuda:0")
    assert (a == check_consisti2v_cli_outputs(base)).all()


@pytest.mark.skipif(not ref_stubs.reference_available() or os.environ.get("ANYV2V_LONG_TESTS", "0") != "1",
                    reason="needs /root/reference; ~2 minutes of fp32 CPU work at 1250 M parameters (ANYV2V_LONG_TESTS=1)")
def test_full_width_unet_fixture_is_what_the_reference_code_produces():
    warnings.filterwarnings("ignore")
    unet_mod, ublocks, pnp = ref_stubs.load_reference_consisti2v_unet()
    ref = spec.fill_weights(unet_mod.VideoLDMUNet3DConditionModel(**spec.unet_full_cfg())).eval()

    def call(u, sample, t, ehs, first, stride):
        with torch.no_grad():
            return u(sample, t, encoder_hidden_states=ehs, first_frame_latents=first, frame_stride=stride).sample
    out = spec.run_unet_full_cases(ref, pnp, call)
    fx = torch.load(os.path.join(HERE, "golden", "consisti2v_unet_full.pt"))
    for k in ("full_nohook", "full_hook_t981"):
        assert float((out[k] - fx[k].float()).abs().max()) <= 2e-3 * float(fx[k].float().abs().max()), k


def test_consisti2v_pipeline_callbacks_and_tensor_first_frames(monkeypatch):
    """``callback(i, t, latents)`` every ``callback_steps`` steps (``pipeline_video_editing.py:697-700``) and ``first_frames`` (an already
    pre-processed frame tensor) instead of ``first_frame_paths`` (``:816-826``)."""
    emu.install(monkeypatch)
    from anyv2v_amd import consisti2v as c2
    from anyv2v_amd.consisti2v_pipeline import ConditionalVideoEditingPipeline, frame_to_pixels
    from anyv2v_amd.schedulers import CONSISTI2V_SCHEDULER_CONFIG, DDIMInverseScheduler
    from hf_clip_reference import HFTextEncoder
    from oracle import ref_consisti2v_pipeline as rcp
    from oracle import ref_pipeline as rp
    j = spec.PIPE_JOB
    frames, _ = spec.pipeline_frames()
    tok = rp.ToyTokenizer()
    pipe = ConditionalVideoEditingPipeline(vae=spec.ToyVaeAdapter(rcp.ToyVAE()), text_encoder=HFTextEncoder(rp.ToyTextEncoder(48), tok), tokenizer=tok,
                                           unet=spec.fill_weights(c2.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)),
                                           scheduler=DDIMInverseScheduler(**CONSISTI2V_SCHEDULER_CONFIG))
    lat0 = pipe.encode_vae_video(frames, pipe.device, height=j["height"], width=j["width"])
    seen = []
    kw = dict(prompt="", height=j["height"], width=j["width"], video_length=j["frames"], num_inference_steps=4, guidance_scale_txt=1.0,
              guidance_scale_img=1.0, negative_prompt="", frame_stride=3, latents=lat0, output_type="latent")
    a = pipe.invert(first_frame_paths=frames[0], callback=lambda i, t, x: seen.append((i, int(t), tuple(x.shape))), callback_steps=2, **kw).videos
    assert [(i, t) for i, t, _ in seen] == [(0, 1), (2, 501)] and seen[0][2] == (1, 4, j["frames"] - 1, j["height"] // 8, j["width"] // 8)
    b = pipe.invert(first_frames=frame_to_pixels(frames[0], j["height"], j["width"], True), **kw).videos
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        pipe.invert(first_frame_paths=frames[0], first_frames=torch.zeros(1, 3, j["height"], j["width"]), **kw)


# ------------------------------------------------------------------------------------------------- FrameInit
@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_frameinit_filters_and_mix_vs_the_references_own_functions():
    """``consisti2v/consisti2v/utils/frameinit_utils.py`` imported verbatim (pure torch) vs ``get_freq_filter`` / ``freq_mix_3d``."""
    import importlib.util
    from anyv2v_amd.consisti2v_pipeline import freq_mix_3d, get_freq_filter
    spec_ = importlib.util.spec_from_file_location("_ref_frameinit", os.path.join(ref_stubs.REFERENCE_ROOT, "consisti2v", "consisti2v", "utils",
                                                                                  "frameinit_utils.py"))
    ref = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(ref)
    shape = [1, 4, 6, 8, 12]
    for kind, n, d_s, d_t in (("gaussian", None, 0.25, 0.25), ("butterworth", 4, 0.25, 0.5), ("butterworth", 2, 0.5, 0.25), ("ideal", None, 0.3, 0.25),
                              ("gaussian", None, 0.0, 0.25)):
        a, b = get_freq_filter(shape, "cpu", kind, n, d_s, d_t), ref.get_freq_filter(shape, "cpu", kind, n, d_s, d_t)
        assert a.shape == b.shape and torch.allclose(a, b, atol=1e-6), kind
    g = torch.Generator().manual_seed(0)
    x, noise = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    lpf = ref.get_freq_filter(shape, "cpu", "butterworth", 4, 0.25, 0.25)
    assert torch.allclose(freq_mix_3d(x, noise, lpf), ref.freq_mix_3d(x, noise, lpf), atol=1e-5)
    box = get_freq_filter(shape, "cpu", "box", None, 0.5, 0.5)       # (the reference's box filter returns None: the mask it builds is checked)
    assert box.sum() == 4 * (2 * round(6 // 2 * 0.5)) * (2 * round(8 // 2 * 0.5)) ** 2 and box[0, 0, 3, 4, 6] == 1


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_native_consisti2v_sampling_with_frameinit_vs_the_reference_pipeline(monkeypatch, tmp_path):
    """``__call__`` from fresh noise with ``use_frameinit`` (``pipeline_video_editing.py:469-711``: pyoco-mixed noise from a seeded generator,
    ``init_filter``, ``add_noise`` at level 999, ``freq_mix_3d``, text guidance) -- the reference's class vs the native pipeline."""
    import types
    warnings.filterwarnings("ignore")
    from anyv2v_amd import consisti2v as c2
    from anyv2v_amd.consisti2v_pipeline import ConditionalVideoEditingPipeline
    from anyv2v_amd.schedulers import CONSISTI2V_SCHEDULER_CONFIG, DDIMScheduler
    from hf_clip_reference import HFTextEncoder
    from oracle import ref_consisti2v_pipeline as rcp
    from oracle import ref_pipeline as rp
    j = spec.PIPE_JOB
    frames, _ = spec.pipeline_frames()
    first = str(tmp_path / "first.png")
    frames[0].resize((j["width"], j["height"])).save(first)
    fp = types.SimpleNamespace(method="butterworth", n=4, d_s=0.25, d_t=0.25)
    kw = dict(prompt="a robot", first_frame_paths=first, height=j["height"], width=j["width"], video_length=j["frames"], num_inference_steps=3,
              guidance_scale_txt=4.0, guidance_scale_img=1.0, negative_prompt="blurry", frame_stride=3, noise_sampling_method="pyoco_mixed",
              use_frameinit=True, frameinit_noise_level=999)
    unet_mod, _, _ = ref_stubs.load_reference_consisti2v_unet()
    ref_unet = spec.fill_weights(unet_mod.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)).eval()
    ref, pm, pnp, inv_mod = rcp.build_reference_pipeline(ref_unet, 48)
    ref.scheduler = rcp.ForwardDDIM(ref.scheduler)
    ref.init_filter(j["frames"], j["height"], j["width"], fp)
    cap = []
    orig = ref.decode_latents
    ref.decode_latents = lambda lat, *a, **k: (cap.append(lat.detach().clone()), orig(lat, *a, **k))[1]
    with torch.no_grad():
        ref(generator=torch.Generator().manual_seed(3), **kw)
    emu.install(monkeypatch)
    tok = rp.ToyTokenizer()
    nat = ConditionalVideoEditingPipeline(vae=spec.ToyVaeAdapter(rcp.ToyVAE()), text_encoder=HFTextEncoder(rp.ToyTextEncoder(48), tok), tokenizer=tok,
                                          unet=spec.fill_weights(c2.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)),
                                          scheduler=DDIMScheduler(**CONSISTI2V_SCHEDULER_CONFIG))
    nat.init_filter(j["frames"], j["height"], j["width"], fp)
    got = nat(generator=torch.Generator().manual_seed(3), output_type="latent", **kw).videos
    ok, err = _close(got, cap[-1], 3e-2)
    assert ok, err
    with pytest.raises(ValueError):
        ConditionalVideoEditingPipeline(vae=nat.vae, text_encoder=nat.text_encoder, tokenizer=tok, unet=nat.unet, scheduler=nat.scheduler)(
            generator=torch.Generator().manual_seed(3), **kw)     # no init_filter


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
@pytest.mark.parametrize("motion", ["pan_left", "pan_right", "zoom_in"])
def test_native_consisti2v_camera_motion_with_frameinit_vs_the_reference_pipeline(monkeypatch, tmp_path, motion):
    """``camera_motion`` (``pipeline_video_editing.py:63-121,553-594``): the pseudo clip cut out of the first frame, its latents as the
    FrameInit layout, its first latent as the conditioning frame -- reference class vs native pipeline, sampling from seeded noise.
    (``zoom_out`` passes float crop sizes to torchvision's crop in the reference and cannot run there.)"""
    import types
    warnings.filterwarnings("ignore")
    from PIL import Image
    from anyv2v_amd import consisti2v as c2
    from anyv2v_amd.consisti2v_pipeline import ConditionalVideoEditingPipeline
    from anyv2v_amd.schedulers import CONSISTI2V_SCHEDULER_CONFIG, DDIMScheduler
    from hf_clip_reference import HFTextEncoder
    from oracle import ref_consisti2v_pipeline as rcp
    from oracle import ref_pipeline as rp
    j = spec.PIPE_JOB
    frames, _ = spec.pipeline_frames()
    first = str(tmp_path / "first.png")
    frames[0].resize((3 * j["width"] // 2, j["height"] + 24), resample=Image.BICUBIC).save(first)     # wider and taller than the target
    fp = types.SimpleNamespace(method="gaussian", n=None, d_s=0.25, d_t=0.25)
    kw = dict(prompt="a robot", first_frame_paths=first, height=j["height"], width=j["height"], video_length=j["frames"], num_inference_steps=2,
              guidance_scale_txt=1.0, guidance_scale_img=1.0, negative_prompt="", frame_stride=3, use_frameinit=True, camera_motion=motion)
    unet_mod, _, _ = ref_stubs.load_reference_consisti2v_unet()
    ref_unet = spec.fill_weights(unet_mod.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)).eval()
    ref, pm, pnp, inv_mod = rcp.build_reference_pipeline(ref_unet, 48)
    ref.scheduler = rcp.ForwardDDIM(ref.scheduler)
    ref.init_filter(j["frames"], j["height"], j["height"], fp)
    cap = []
    orig = ref.decode_latents
    ref.decode_latents = lambda lat, *a, **k: (cap.append(lat.detach().clone()), orig(lat, *a, **k))[1]
    with torch.no_grad():
        ref(generator=torch.Generator().manual_seed(3), **kw)
    emu.install(monkeypatch)
    tok = rp.ToyTokenizer()
    nat = ConditionalVideoEditingPipeline(vae=spec.ToyVaeAdapter(rcp.ToyVAE()), text_encoder=HFTextEncoder(rp.ToyTextEncoder(48), tok), tokenizer=tok,
                                          unet=spec.fill_weights(c2.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)),
                                          scheduler=DDIMScheduler(**CONSISTI2V_SCHEDULER_CONFIG))
    nat.init_filter(j["frames"], j["height"], j["height"], fp)
    got = nat(generator=torch.Generator().manual_seed(3), output_type="latent", **kw).videos
    ok, err = _close(got, cap[-1], 2e-2)
    assert ok, err


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
@pytest.mark.parametrize("cls", ["ConditionalAnimationPipeline", "AutoregressiveAnimationPipeline"])
def test_native_consisti2v_animation_pipelines_vs_the_references_own_classes(monkeypatch, tmp_path, cls):
    """ConsistI2V's two samplers next to the editing pipeline (``pipeline_conditional_animation.py:462-703``,
    ``pipeline_autoregress_animation.py:401-615``), their files imported verbatim: a first frame of ANOTHER aspect ratio (Resize(height) +
    CenterCrop), text + image guidance (three rows, the image-unconditional one on the noisy first frame), FrameInit, pyoco noise; the
    autoregressive one two chunks long (the second conditioned on the last latent frame of the first, fresh noise of the same generator)."""
    import types
    warnings.filterwarnings("ignore")
    from PIL import Image
    from anyv2v_amd import consisti2v as c2
    from anyv2v_amd import consisti2v_pipeline as cp
    from anyv2v_amd.schedulers import CONSISTI2V_SCHEDULER_CONFIG, DDIMScheduler
    from hf_clip_reference import HFTextEncoder
    from oracle import ref_consisti2v_pipeline as rcp
    from oracle import ref_pipeline as rp
    j = spec.PIPE_JOB
    frames, _ = spec.pipeline_frames()
    first = str(tmp_path / "first.png")
    frames[0].resize((j["width"] + 40, j["height"] + 8), resample=Image.BICUBIC).save(first)
    fp = types.SimpleNamespace(method="butterworth", n=4, d_s=0.25, d_t=0.25)
    kw = dict(prompt="a robot", first_frame_paths=first, height=j["height"], width=j["width"], video_length=j["frames"], num_inference_steps=2,
              guidance_scale_txt=3.0, guidance_scale_img=1.5, negative_prompt="blurry", frame_stride=3, noise_sampling_method="pyoco_progressive",
              noise_alpha=0.7, use_frameinit=True, frameinit_noise_level=900)
    n_frames = j["frames"]
    if cls == "AutoregressiveAnimationPipeline":
        kw["autoregress_steps"] = 2
        n_frames = 2 * j["frames"] - 1
    unet_mod, _, _ = ref_stubs.load_reference_consisti2v_unet()
    ref_unet = spec.fill_weights(unet_mod.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)).eval()
    ref, pm, pnp, inv_mod = rcp.build_reference_pipeline(ref_unet, 48, cls)
    assert type(ref).__name__ == cls and not hasattr(ref, "invert")
    ref.scheduler = rcp.ForwardDDIM(ref.scheduler)
    ref.init_filter(j["frames"], j["height"], j["width"], fp)
    cap = []
    orig = ref.decode_latents
    ref.decode_latents = lambda lat, *a, **k: (cap.append(lat.detach().clone()), orig(lat, *a, **k))[1]
    with torch.no_grad():
        ref_video = ref(generator=torch.Generator().manual_seed(5), **kw).videos
    assert cap[-1].shape[2] == n_frames
    emu.install(monkeypatch)
    tok = rp.ToyTokenizer()
    nat = getattr(cp, cls)(vae=spec.ToyVaeAdapter(rcp.ToyVAE()), text_encoder=HFTextEncoder(rp.ToyTextEncoder(48), tok), tokenizer=tok,
                           unet=spec.fill_weights(c2.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)),
                           scheduler=DDIMScheduler(**CONSISTI2V_SCHEDULER_CONFIG))
    nat.init_filter(j["frames"], j["height"], j["width"], fp)
    got = nat(generator=torch.Generator().manual_seed(5), output_type="latent", **kw).videos
    assert got.shape == cap[-1].shape
    ok, err = _close(got[:, :, 0], cap[-1][:, :, 0], 2e-3)      # the cropped first frame's latent
    assert ok, err
    ok, err = _close(got, cap[-1], 3e-2)
    assert ok, err
    video = nat(generator=torch.Generator().manual_seed(5), **kw).videos
    assert tuple(video.shape) == tuple(ref_video.shape)
    ok, err = _close(video, ref_video, 3e-2)
    assert ok, err
    with pytest.raises(TypeError):
        nat(generator=torch.Generator().manual_seed(5), ddim_init_latents_t_idx=1, **kw)
    with pytest.raises(AttributeError):
        nat.invert(prompt="", video_length=j["frames"])


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
@pytest.mark.parametrize("eta", [0.0, 0.5])
def test_native_consisti2v_guidance_rescale_and_eta_vs_the_reference_pipeline(monkeypatch, tmp_path, eta):
    """``guidance_rescale`` (``pipeline_video_editing.py:50-61,685-688``) and ``eta`` (``:373-388`` -> the scheduler's step; the variance
    noise from the call's generator, after the start noise) through ``__call__`` under text guidance -- reference class vs native."""
    warnings.filterwarnings("ignore")
    from anyv2v_amd import consisti2v as c2
    from anyv2v_amd.consisti2v_pipeline import ConditionalVideoEditingPipeline
    from anyv2v_amd.schedulers import CONSISTI2V_SCHEDULER_CONFIG, DDIMScheduler
    from hf_clip_reference import HFTextEncoder
    from oracle import ref_consisti2v_pipeline as rcp
    from oracle import ref_pipeline as rp
    j = spec.PIPE_JOB
    frames, _ = spec.pipeline_frames()
    first = str(tmp_path / "first.png")
    frames[0].resize((j["width"], j["height"])).save(first)
    kw = dict(prompt="a robot", first_frame_paths=first, height=j["height"], width=j["width"], video_length=j["frames"], num_inference_steps=3,
              guidance_scale_txt=6.0, guidance_scale_img=1.0, negative_prompt="blurry", frame_stride=3, guidance_rescale=0.7, eta=eta)
    unet_mod, _, _ = ref_stubs.load_reference_consisti2v_unet()
    ref_unet = spec.fill_weights(unet_mod.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)).eval()
    ref, pm, pnp, inv_mod = rcp.build_reference_pipeline(ref_unet, 48)
    ref.scheduler = rcp.ForwardDDIM(ref.scheduler)
    cap = []
    orig = ref.decode_latents
    ref.decode_latents = lambda lat, *a, **k: (cap.append(lat.detach().clone()), orig(lat, *a, **k))[1]
    with torch.no_grad():
        ref(generator=torch.Generator().manual_seed(3), **kw)
        plain = dict(kw, guidance_rescale=0.0, eta=0.0)
        ref(generator=torch.Generator().manual_seed(3), **plain)
    assert (cap[0] - cap[1]).abs().max() > 0.05                 # the options do something at this size
    emu.install(monkeypatch)
    tok = rp.ToyTokenizer()
    nat = ConditionalVideoEditingPipeline(vae=spec.ToyVaeAdapter(rcp.ToyVAE()), text_encoder=HFTextEncoder(rp.ToyTextEncoder(48), tok), tokenizer=tok,
                                          unet=spec.fill_weights(c2.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)),
                                          scheduler=DDIMScheduler(**CONSISTI2V_SCHEDULER_CONFIG))
    got = nat(generator=torch.Generator().manual_seed(3), output_type="latent", **kw).videos
    ok, err = _close(got, cap[0], 3e-2)
    assert ok, err
    with pytest.raises(NotImplementedError):                    # (the reference raises NameError there)
        nat(generator=torch.Generator().manual_seed(3), output_type="latent", **dict(kw, guidance_scale_img=2.0))


def test_ddim_scheduler_step_with_eta_draws_from_the_generator(monkeypatch):
    """``DDIMScheduler.step(eta=)``: the coefficients of ``eta_coefficients`` + variance noise from ``generator`` (or ``variance_noise``);
    the inverse scheduler refuses eta."""
    from anyv2v_amd.schedulers import CONSISTI2V_SCHEDULER_CONFIG, DDIMInverseScheduler, DDIMScheduler
    emu.install(monkeypatch)
    sched = DDIMScheduler(**CONSISTI2V_SCHEDULER_CONFIG)
    sched.set_timesteps(10)
    g = torch.Generator().manual_seed(1)
    x, e = torch.randn(1, 4, 3, 4, 4, generator=g).half(), torch.randn(1, 4, 3, 4, 4, generator=g).half()
    t = int(sched.timesteps[3])
    sa_t, sb_t, cx, ce, sigma = sched.eta_coefficients(t, 0.8)
    assert sigma > 0 and abs(cx ** 2 + ce ** 2 + sigma ** 2 - 1.0) < 1e-6
    n = torch.randn(e.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float32)
    want = cx * (x.float() - sb_t * e.float()) / sa_t + ce * e.float() + sigma * n.half().float()
    got = sched.step(e, t, x, eta=0.8, generator=torch.Generator().manual_seed(5)).prev_sample
    assert torch.allclose(got.float(), want, atol=4e-3)
    got2 = sched.step(e, t, x, eta=0.8, variance_noise=n).prev_sample
    assert torch.equal(got, got2)
    assert torch.equal(sched.step(e, t, x).prev_sample, sched.step(e, t, x, eta=0.0).prev_sample)
    inv = DDIMInverseScheduler(**CONSISTI2V_SCHEDULER_CONFIG)
    inv.set_timesteps(10)
    with pytest.raises(TypeError):          # (the vendored inverse scheduler's step has no eta)
        inv.step(e, int(inv.timesteps[2]), x, eta=0.5)


SAMPLING_TOL = 3e-2


def test_native_consisti2v_sampling_cases_vs_reference_fixture(monkeypatch):
    """``tests/golden/consisti2v_sampling.pt`` (the reference's three pipeline classes sampling from seeded noise: both animation
    pipelines, ``guidance_rescale`` + ``eta``) vs the native pipelines on the emulated kernels; the GPU run compares with the same file."""
    warnings.filterwarnings("ignore")
    fx = torch.load(os.path.join(os.path.dirname(__file__), "golden", "consisti2v_sampling.pt"))
    assert fx["spec"]["cases"] == {k: [c, dict(kw)] for k, (c, kw) in spec.sampling_cases().items()} and fx["spec"]["seed"] == spec.SAMPLING_SEED
    emu.install(monkeypatch)
    got = spec.native_sampling("cpu")
    assert set(got) == set(spec.sampling_cases())
    for name, lat in got.items():
        assert lat.shape == fx[name].shape, name
        ok, err = _close(lat, fx[name], SAMPLING_TOL)
        assert ok, (name, err)


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_sampling_fixture_is_what_the_reference_classes_produce(tmp_path):
    from oracle import ref_consisti2v_pipeline as rcp
    warnings.filterwarnings("ignore")
    fx = torch.load(os.path.join(os.path.dirname(__file__), "golden", "consisti2v_sampling.pt"))
    cases = {k: v for k, v in spec.sampling_cases().items() if k != "animation"}     # (the third is re-run by the class-level test above)
    got = rcp.run_reference_sampling(spec.UNET_CFG, spec.fill_weights, cases, spec.sampling_first_frame(), spec.SAMPLING_FILTER,
                                     spec.SAMPLING_SEED, tmp_path)
    for name, lat in got.items():
        assert torch.equal(lat.half(), fx[name]), name


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
@pytest.mark.parametrize("hw", [(9, 10), (12, 12)])
def test_native_consisti2v_unet_at_latent_sizes_that_are_not_multiples_of_8_vs_the_references_class(monkeypatch, hw):
    """``videoldm_unet.py:726-734,990-1010`` (``forward_upsample_size``): the reference's own UNet class vs the native one at sizes that
    three ceil-halvings do not give back by doubling."""
    warnings.filterwarnings("ignore")
    from anyv2v_amd import consisti2v as c2
    unet_mod, _, _ = ref_stubs.load_reference_consisti2v_unet()
    ref = spec.fill_weights(unet_mod.VideoLDMUNet3DConditionModel(**spec.UNET_CFG)).eval()
    emu.install(monkeypatch)
    nat = spec.fill_weights(c2.VideoLDMUNet3DConditionModel(**spec.UNET_CFG))
    F = spec.UNET_CFG["n_frames"]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, F - 1, *hw, generator=g).half().float()
    ehs = torch.randn(2, 5, spec.UNET_CFG["cross_attention_dim"], generator=g).half().float()
    ff = torch.randn(2, 4, 1, *hw, generator=g).half().float()
    with torch.no_grad():
        want = ref(x, 981, encoder_hidden_states=ehs, first_frame_latents=ff, frame_stride=3).sample
        got = nat(x.half(), 981, encoder_hidden_states=ehs.half(), first_frame_latents=ff.half(), frame_stride=3).sample
    assert got.shape == want.shape
    err = float((got.float() - want).abs().max() / want.abs().max())
    assert err < 8e-3, err


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_shipped_configs_resolve_to_the_references_values():
    """``configs/consisti2v/pipeline_{256,512}/*.yaml`` are laid out differently from the reference's files but hold the same keys and
    values; the I2VGen-XL group templates likewise, except ``device`` (the reference pins GPUs 7 / 4 of its own node)."""
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    load = lambda p: yaml.safe_load(open(p))
    for size in (256, 512):
        for name in (f"ddim_inversion_{size}.yaml", "pnp_edit.yaml"):
            mine = load(os.path.join(root, "configs", "consisti2v", f"pipeline_{size}", name))
            ref = load(os.path.join(ref_stubs.REFERENCE_ROOT, "consisti2v", "configs", f"pipeline_{size}", name))
            assert mine == ref, (size, name)
    for stage in ("group_ddim_inversion", "group_pnp_edit"):
        mine = load(os.path.join(root, "configs", stage, "template.yaml"))
        ref = load(os.path.join(ref_stubs.REFERENCE_ROOT, "i2vgen-xl", "configs", stage, "template.yaml"))
        assert mine.pop("device") == "cuda:0" and ref.pop("device").startswith("cuda:")
        assert mine == ref, stage


def test_camera_motion_zoom_out_cuts_growing_windows():
    from anyv2v_amd.consisti2v_pipeline import camera_motion_frames
    x = torch.arange(3 * 90 * 120, dtype=torch.float32).view(3, 90, 120) / 1000
    z = camera_motion_frames(x, "zoom_out", 4, 32)
    assert tuple(z.shape) == (4, 3, 32, 32)
    # window sizes 60, 67, 75, 82 around the centre: the corner pixel moves outwards
    assert z[0, 0, 0, 0] > z[1, 0, 0, 0] > z[3, 0, 0, 0]


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_epsilon_forward_ddim_step_is_pinned_to_the_references_gaussian_diffusion():
    """The forward DDIM step of this backend (epsilon prediction, linear betas) -- the product's ``DDIMScheduler.coefficients`` and the
    stand-in the reference pipeline harness steps with (``oracle.ref_consisti2v_pipeline.ForwardDDIM``) -- against the in-tree
    ``seine/diffusion/gaussian_diffusion.py::ddim_sample`` on ``SpacedDiffusion`` (eta 0), 50 steps with ``steps_offset`` 1."""
    import importlib.util
    import sys
    from anyv2v_amd.schedulers import CONSISTI2V_SCHEDULER_CONFIG, DDIMScheduler
    from oracle import ref_consisti2v_pipeline as rcp
    root = os.path.join(ref_stubs.REFERENCE_ROOT, "seine", "diffusion")
    spec_ = importlib.util.spec_from_file_location("_ref_seine_diffusion3", os.path.join(root, "__init__.py"), submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec_)
    sys.modules["_ref_seine_diffusion3"] = mod
    try:
        spec_.loader.exec_module(mod)
        gd = sys.modules["_ref_seine_diffusion3.gaussian_diffusion"]
        sched = DDIMScheduler(**CONSISTI2V_SCHEDULER_CONFIG)
        sched.set_timesteps(50)
        ts = sorted(int(t) for t in sched.timesteps)
        diff = mod.SpacedDiffusion(use_timesteps=ts, betas=sched.betas.double().numpy(), model_mean_type=gd.ModelMeanType.EPSILON,
                                   model_var_type=gd.ModelVarType.FIXED_SMALL, loss_type=gd.LossType.MSE)
        inv = ref_stubs.load_reference_inverse_scheduler().DDIMInverseScheduler(**rcp.SCHED_CFG)
        fwd = rcp.ForwardDDIM(inv)
        fwd.set_timesteps(50)
        g = torch.Generator().manual_seed(2)
        x, e = torch.randn(2, 4, 3, 5, 5, generator=g, dtype=torch.float64), torch.randn(2, 4, 3, 5, 5, generator=g, dtype=torch.float64)
        for i, t in enumerate(ts):
            if i not in (0, 1, 10, 30, 49):
                continue
            ref = diff.ddim_sample(lambda xx, tt: e, x, torch.tensor([i, i]), clip_denoised=False, eta=0.0)["sample"]
            sa_t, sb_t, sa_p, sb_p = sched.coefficients(t)
            prod = sa_p * (x - sb_t * e) / sa_t + sb_p * e
            assert torch.allclose(prod, ref, rtol=2e-4, atol=2e-5), ("product", t)
            assert torch.allclose(fwd.step(e, t, x).prev_sample, ref, rtol=2e-4, atol=2e-5), ("harness stand-in", t)
            # eta > 0 (``gaussian_diffusion.py:583-599``; its noise is ``randn_like`` of the global RNG: same seed, same draw)
            torch.manual_seed(40 + i)
            ref = diff.ddim_sample(lambda xx, tt: e, x, torch.tensor([i, i]), clip_denoised=False, eta=0.6)["sample"]
            torch.manual_seed(40 + i)
            noise = torch.randn_like(x)
            sa_t, sb_t, cx, ce, sigma = sched.eta_coefficients(t, 0.6)
            prod = cx * (x - sb_t * e) / sa_t + ce * e + (sigma * noise if i else 0.0)
            assert torch.allclose(prod, ref, rtol=2e-4, atol=2e-5), ("product, eta", t)
            if i:       # (at the last step diffusers still adds s * n; s = 0 there with alpha_prev = 1, not with this family's final alpha)
                g2 = torch.Generator().manual_seed(9)
                n2 = torch.randn(e.shape, generator=torch.Generator().manual_seed(9), dtype=torch.float32).double()
                want = cx * (x - sb_t * e) / sa_t + ce * e + sigma * n2
                assert torch.allclose(fwd.step(e, t, x, eta=0.6, generator=g2).prev_sample, want, rtol=2e-4, atol=2e-5), ("harness stand-in, eta", t)
    finally:
        for k in [k for k in sys.modules if k.startswith("_ref_seine_diffusion3")]:
            del sys.modules[k]
