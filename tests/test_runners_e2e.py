"""End-to-end CLI runners on CPU (TEST-ONLY op emulation): stage 1 ``run_group_ddim_inversion`` -> stage 2
``run_group_pnp_edit`` on a tiny synthetic clip with a mini UNet checkpoint (``unet/config.json`` + safetensors in diffusers
key naming) and the weight-free stand-in encoders.  Checks the on-disk formats the reference uses on either side of the hot
path (``ddim_latents_{t}.pt``, PNG frame dirs, GIF) and that ``--frame_parallel`` under a world_size-2
gloo group (F3) reproduces the single-process outputs with rank 0 as the only writer."""
import json
import logging
import os
import socket
import sys

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES, SIZE, N_STEPS = 4, 128, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_workspace(base):
    """<base>/model (mini checkpoint), <base>/demo/<clip> (PNG frames + edited first frame), configs."""
    sys.path.insert(0, ROOT)
    from safetensors.torch import save_file
    from oracle.unet_oracle import UNetConfig, random_state_dict
    base = str(base)
    os.makedirs(os.path.join(base, "model", "unet"), exist_ok=True)
    sd = {k: v.half().contiguous() for k, v in random_state_dict(UNetConfig.mini(), 1234).items()}
    save_file(sd, os.path.join(base, "model", "unet", "diffusion_pytorch_model.fp16.safetensors"))
    json.dump({"_class_name": "I2VGenXLUNet", "block_out_channels": [64, 128, 256, 256], "cross_attention_dim": 128,
               "attention_head_dim": 64, "in_channels": 4, "out_channels": 4, "layers_per_block": 2, "norm_num_groups": 32,
               "sample_size": 8, "transformer_in_heads": 2}, open(os.path.join(base, "model", "unet", "config.json"), "w"))
    clip = os.path.join(base, "demo", "clip")
    os.makedirs(os.path.join(clip, "edited_first_frame"), exist_ok=True)
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:SIZE, 0:SIZE]
    for i in range(N_FRAMES):
        img = np.stack([(xx * 2 + 10 * i) % 256, (yy * 2) % 256, ((xx + yy) + 30 * i) % 256], -1).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(clip, f"{i:05d}.png"))
    Image.fromarray(rng.integers(0, 255, (SIZE, SIZE, 3), dtype=np.uint8)).save(os.path.join(clip, "edited_first_frame", "e.png"))
    return base


def _configs(base, tag):
    from anyv2v_amd.config import OmegaConf
    inv = OmegaConf.load(os.path.join(ROOT, "configs", "group_ddim_inversion", "template.yaml"))
    ed = OmegaConf.load(os.path.join(ROOT, "configs", "group_pnp_edit", "template.yaml"))
    for c in (inv, ed):
        c.device, c.data_dir, c.model_path = "cpu", base, os.path.join(base, "model")
        c.image_size, c.n_frames, c.model_name = [SIZE, SIZE], N_FRAMES, f"mini-{tag}"
    inv.inverse_config.n_steps = N_STEPS
    inv.recon_config.n_steps = N_STEPS
    inv.recon_config.ddim_init_latents_t_idx = 0
    ed.n_steps, ed.ddim_init_latents_t_idx = N_STEPS, 0
    inv_list = [{"active": True, "force_recompute_latents": False, "video_name": "clip", "recon_config": {"enable_recon": True}},
                {"active": False, "video_name": "unused"}]
    ed_list = [{"active": True, "task_name": "Prompt-Based-Editing", "video_name": "clip",
                "edited_first_frame_path": "demo/clip/edited_first_frame/e.png", "editing_prompt": "a robot",
                "edited_video_name": "robot", "ddim_init_latents_t_idx": 0, "pnp_f_t": 0.25, "pnp_spatial_attn_t": 0.5,
                "pnp_temp_attn_t": 0.75}]
    return inv, inv_list, ed, ed_list


def _run_both_stages(base, tag, frame_parallel=False, device="cpu"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if device == "cpu":  # TEST-ONLY emulation of the C-ABI ops; on a GPU the runners use the HIP library and HIP graphs
        import cpu_ops_emulation as emu
        emu.install()
        os.environ["ANYV2V_NO_GRAPH"] = "1"
    torch.set_grad_enabled(False)
    from anyv2v_amd import run_group_ddim_inversion as s1, run_group_pnp_edit as s2
    from anyv2v_amd.utils import seed_everything
    inv, inv_list, ed, ed_list = _configs(base, tag)
    inv.device = ed.device = device
    log = logging.getLogger("e2e")
    dev = torch.device(device)
    seed_everything(inv.seed)
    s1.main(inv, inv_list, dev, log, synthetic_encoders=True, frame_parallel=frame_parallel)
    seed_everything(ed.seed)
    s2.main(ed, ed_list, dev, log, synthetic_encoders=True, frame_parallel=frame_parallel)


def _outputs(base, tag):
    inv_dir = os.path.join(base, "inversions", f"mini-{tag}", "clip")
    res_root = os.path.join(base, "Results", "Prompt-Based-Editing", f"mini-{tag}", "clip", "robot")
    sub = os.listdir(res_root)
    assert len(sub) == 1
    return inv_dir, os.path.join(res_root, sub[0])


def test_stage1_stage2_file_formats(tmp_path):
    _check_file_formats(tmp_path, "cpu")


def _check_front_end_callable(tmp_path, device):
    """``anyv2v_amd.api.AnyV2V_I2VGenXL.perform_anyv2v`` -- the function behind ``gradio_demo.py:80-222`` / ``predict.py``: an
    mp4 clip + an edited first frame in, ``edited_video.mp4`` out, the trajectory files on disk, deterministic in the seed, and
    the same edited frames as assembling the steps by hand the way the reference's demo does (files read back from disk)."""
    base = _make_workspace(tmp_path)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if device == "cpu":  # TEST-ONLY emulation of the C-ABI ops; on a GPU: the HIP library and HIP graphs
        import cpu_ops_emulation as emu
        emu.install()
        os.environ["ANYV2V_NO_GRAPH"] = "1"
    torch.set_grad_enabled(False)
    from anyv2v_amd.api import AnyV2V_I2VGenXL
    from anyv2v_amd.mp4 import read_mp4
    from anyv2v_amd.utils import export_to_video
    clip = os.path.join(base, "demo", "clip")
    frames = [Image.open(os.path.join(clip, f"{i:05d}.png")).convert("RGB") for i in range(N_FRAMES)]
    src = export_to_video(frames, os.path.join(base, "clip.mp4"), fps=8)
    ed = AnyV2V_I2VGenXL(model_path=os.path.join(base, "model"), device=device, tmp_dir=os.path.join(base, "tmp"),
                         synthetic_encoders=True)
    args = dict(video_path=src, video_prompt="a robot", video_negative_prompt="blurry",
                edited_first_frame_path=os.path.join(clip, "edited_first_frame", "e.png"), conv_inj=0.25, spatial_inj=0.5,
                temp_inj=0.75, num_inference_steps=N_STEPS, guidance_scale=9.0, ddim_init_latents_t_idx=0,
                ddim_inversion_steps=N_STEPS, seed=7)
    out = ed.perform_anyv2v(**args)
    assert out.endswith(os.path.join("AnyV2V", "edited_video.mp4"))
    vid, fps = read_mp4(out)
    assert len(vid) == N_FRAMES and vid[0].size == (SIZE, SIZE) and fps == 8.0
    lat_dir = os.path.join(base, "tmp", "AnyV2V", "ddim_latents")
    assert len([f for f in os.listdir(lat_dir) if f.startswith("ddim_latents_")]) == N_STEPS
    first = [np.asarray(f).copy() for f in vid]
    # by hand, as gradio_demo.py does: start latent and source trajectory read back from the files
    from anyv2v_amd.run_group_pnp_edit import init_pnp
    from anyv2v_amd.utils import load_ddim_latents_at_t, load_image
    pipe, sched = ed.pipe, ed.ddim_scheduler
    sched.set_timesteps(N_STEPS)
    lat_t = load_ddim_latents_at_t(sched.timesteps[0], ddim_latents_path=lat_dir)
    g = torch.Generator(device=device).manual_seed(7)
    # the generator state after the inversion of the first call: replay its draws (encode_vae_video samples the posterior)
    from anyv2v_amd.run_group_ddim_inversion import ddim_inversion
    cfg = ed.config.inverse_config
    cfg.output_dir = os.path.join(base, "tmp", "by_hand")
    src_frames = read_mp4(src)[0]
    ddim_inversion(cfg, src_frames[0], src_frames, pipe, ed.inverse_scheduler, g)
    pipe._last_trajectory.wait()
    torch.randn_like(lat_t.to(device))  # (random_ratio = 0: the blend draws and discards, gradio_demo.py:160)
    init_pnp(pipe, sched, ed.config.pnp_config)
    pipe.register_modules(scheduler=sched)
    e1 = load_image(args["edited_first_frame_path"]).resize((SIZE, SIZE), resample=Image.Resampling.LANCZOS)
    by_hand = pipe.sample_with_pnp(prompt="a robot", image=e1, height=SIZE, width=SIZE, num_frames=N_FRAMES,
                                   num_inference_steps=N_STEPS, guidance_scale=9.0, negative_prompt="blurry", target_fps=8,
                                   latents=lat_t.to(device), generator=g, return_dict=True, ddim_init_latents_t_idx=0,
                                   ddim_inv_latents_path=cfg.output_dir, ddim_inv_prompt="", ddim_inv_1st_frame=src_frames[0]).frames[0]
    export_to_video(by_hand, os.path.join(base, "by_hand.mp4"), fps=8)
    hand = [np.asarray(f) for f in read_mp4(os.path.join(base, "by_hand.mp4"))[0]]
    assert all(np.array_equal(a, b) for a, b in zip(first, hand)), "HBM trajectory hand-over != files read back"
    # same seed -> same video
    out2 = ed.perform_anyv2v(**args)
    assert all(np.array_equal(a, np.asarray(b)) for a, b in zip(first, read_mp4(out2)[0]))


def test_front_end_callable_perform_anyv2v(tmp_path):
    _check_front_end_callable(tmp_path, "cpu")


def test_front_end_takes_a_clip_whose_size_is_not_a_multiple_of_64(tmp_path):
    """``gradio_demo.py:129`` / ``predict.py:153``: the front ends run a clip at ITS size.  160 x 88 pixels are 20 x 11 latents; 11 -> 6 -> 3 -> 2
    does not come back by doubling, so the UNet resizes to the skip connections' sizes on the way up (diffusers ``forward_upsample_size``)."""
    base = _make_workspace(tmp_path)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ops_emulation as emu
    emu.install()
    os.environ["ANYV2V_NO_GRAPH"] = "1"
    torch.set_grad_enabled(False)
    from anyv2v_amd.api import AnyV2V_I2VGenXL
    from anyv2v_amd.mp4 import read_mp4
    from anyv2v_amd.utils import export_to_video
    W, H = 160, 88
    yy, xx = np.mgrid[0:H, 0:W]
    frames = [Image.fromarray(np.stack([(xx * 2 + 10 * i) % 256, (yy * 3) % 256, ((xx + yy) + 30 * i) % 256], -1).astype(np.uint8))
              for i in range(N_FRAMES)]
    src = export_to_video(frames, os.path.join(base, "wide.mp4"), fps=8)
    ed = AnyV2V_I2VGenXL(model_path=os.path.join(base, "model"), device="cpu", tmp_dir=os.path.join(base, "tmp"), synthetic_encoders=True)
    args = dict(video_path=src, video_prompt="a robot", video_negative_prompt="blurry",
                edited_first_frame_path=os.path.join(base, "demo", "clip", "edited_first_frame", "e.png"), conv_inj=0.25, spatial_inj=0.5,
                temp_inj=0.75, num_inference_steps=N_STEPS, guidance_scale=9.0, ddim_init_latents_t_idx=0, ddim_inversion_steps=N_STEPS, seed=7)
    vid, fps = read_mp4(ed.perform_anyv2v(**args))
    assert len(vid) == N_FRAMES and vid[0].size == (W, H) and fps == 8.0
    a = np.stack([np.asarray(f) for f in vid]).astype(np.float32)
    assert np.isfinite(a).all() and a.std() > 5.0
    lat = torch.load(os.path.join(base, "tmp", "AnyV2V", "ddim_latents", os.listdir(os.path.join(base, "tmp", "AnyV2V", "ddim_latents"))[0]))
    assert tuple(lat.shape[-2:]) == (H // 8, W // 8)
    vid2, _ = read_mp4(ed.perform_anyv2v(**args))       # same seed -> same video
    assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(vid, vid2))


@pytest.mark.gpu
def test_front_end_callable_on_gpu(tmp_path):
    assert torch.cuda.is_available()
    _check_front_end_callable(tmp_path, "cuda")


def test_stage1_from_an_mp4_clip(tmp_path):
    """The reference's fallback when ``<video_dir>/<name>/%05d.png`` is missing (``run_group_ddim_inversion.py:97-105``):
    decode ``<video_dir>/<name>.mp4`` into that directory, then go on as usual."""
    base = _make_workspace(tmp_path)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ops_emulation as emu
    emu.install()
    os.environ["ANYV2V_NO_GRAPH"] = "1"
    torch.set_grad_enabled(False)
    from anyv2v_amd import run_group_ddim_inversion as s1
    from anyv2v_amd.utils import export_to_video, seed_everything
    clip = os.path.join(base, "demo", "clip")
    pngs = [os.path.join(clip, f"{i:05d}.png") for i in range(N_FRAMES)]
    export_to_video([Image.open(p).convert("RGB") for p in pngs], os.path.join(base, "demo", "clip.mp4"), fps=8)
    for p in pngs:
        os.remove(p)
    inv, inv_list, _, _ = _configs(base, "mp4")
    inv_list[0]["recon_config"] = {"enable_recon": False}
    seed_everything(inv.seed)
    s1.main(inv, inv_list, torch.device("cpu"), logging.getLogger("e2e"), synthetic_encoders=True)
    assert all(os.path.isfile(p) for p in pngs) and os.path.isfile(os.path.join(clip, "clip.gif"))
    lat_dir = os.path.join(base, "inversions", "mini-mp4", "clip", "ddim_latents")
    assert len(os.listdir(lat_dir)) == N_STEPS


@pytest.mark.gpu
def test_stage1_stage2_on_gpu(tmp_path):
    """The same two CLI stages on cuda:0 through the HIP library (HIP-graph step engines, real kernels)."""
    assert torch.cuda.is_available()
    _check_file_formats(tmp_path, "cuda")


def _check_file_formats(tmp_path, device):
    base = _make_workspace(tmp_path)
    _run_both_stages(base, "single", device=device)
    inv_dir, out_dir = _outputs(base, "single")
    lat_files = sorted(os.listdir(os.path.join(inv_dir, "ddim_latents")))
    assert len(lat_files) == N_STEPS and all(f.startswith("ddim_latents_") and f.endswith(".pt") for f in lat_files)
    x = torch.load(os.path.join(inv_dir, "ddim_latents", lat_files[0]))
    assert tuple(x.shape) == (1, 4, N_FRAMES, SIZE // 8, SIZE // 8) and torch.isfinite(x.float()).all()
    # reconstruction: 512 x 512 whatever the clip's size, mp4 at 10 fps and a GIF at diffusers' fixed 100 ms per frame (:181-191)
    from anyv2v_amd.mp4 import read_mp4
    rec, rec_fps = read_mp4(os.path.join(inv_dir, "ddim_reconstruction.mp4"))
    assert len(rec) == N_FRAMES and rec[0].size == (512, 512) and rec_fps == 10.0
    gif = Image.open(os.path.join(inv_dir, "ddim_reconstruction.gif"))
    assert gif.n_frames == N_FRAMES and gif.size == (512, 512) and gif.info["duration"] == 100 and gif.info.get("loop") == 0
    assert "ddim_init_latents_t_idx_0_nsteps_4_cfg_9.0_pnpf0.25_pnps0.5_pnpt0.75" == os.path.basename(out_dir)
    names = sorted(os.listdir(out_dir))
    assert "video.gif" in names and "video.mp4" in names and "edited_latents.pt" in names
    assert Image.open(os.path.join(out_dir, "video.gif")).info["duration"] == 100        # (no rate given at run_group_pnp_edit.py:179)
    assert read_mp4(os.path.join(out_dir, "video.mp4"))[1] == 8.0                         # (fps=config.target_fps at :178)
    assert [n for n in names if n.endswith(".png")] == [f"video_{i:05d}.png" for i in range(N_FRAMES)]
    lat = torch.load(os.path.join(out_dir, "edited_latents.pt"))
    assert tuple(lat.shape) == (1, 4, N_FRAMES, SIZE // 8, SIZE // 8) and torch.isfinite(lat.float()).all()
    with Image.open(os.path.join(out_dir, "video.gif")) as g:
        assert g.n_frames == N_FRAMES and g.size == (SIZE, SIZE)
    # the reference's export_to_video(..., "video.mp4", fps=target_fps) (run_group_pnp_edit.py:178): an H.264 mp4 with the frames
    from anyv2v_amd.mp4 import read_mp4
    import numpy as np
    vid, fps = read_mp4(os.path.join(out_dir, "video.mp4"))
    assert len(vid) == N_FRAMES and vid[0].size == (SIZE, SIZE) and fps > 0
    png0 = np.asarray(Image.open(os.path.join(out_dir, "video_00000.png")).convert("RGB"), dtype=np.int16)
    assert np.abs(np.asarray(vid[0], dtype=np.int16) - png0).mean() < 6.0  # 4:2:0 chroma subsampling + limited-range rounding
    # stage 1 is skipped when its output exists (reference behaviour, run_group_ddim_inversion.py:118-120)
    mtime = os.path.getmtime(os.path.join(inv_dir, "ddim_latents", lat_files[0]))
    _run_both_stages(base, "single", device=device)
    assert os.path.getmtime(os.path.join(inv_dir, "ddim_latents", lat_files[0])) == mtime


def _fp_worker(rank, world, port, base):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    _run_both_stages(base, "fp", frame_parallel=True)
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_frame_parallel_runners_world2_match_single_process(tmp_path):
    import torch.multiprocessing as mp
    base = _make_workspace(tmp_path)
    _run_both_stages(base, "single")
    mp.spawn(_fp_worker, args=(2, _free_port(), base), nprocs=2, join=True)
    (inv_a, out_a), (inv_b, out_b) = _outputs(base, "single"), _outputs(base, "fp")
    fa, fb = sorted(os.listdir(os.path.join(inv_a, "ddim_latents"))), sorted(os.listdir(os.path.join(inv_b, "ddim_latents")))
    assert fa == fb and len(fa) == N_STEPS
    for f in fa:
        a, b = torch.load(os.path.join(inv_a, "ddim_latents", f)).float(), torch.load(os.path.join(inv_b, "ddim_latents", f)).float()
        assert (a - b).abs().max() <= 2e-2 * a.abs().max(), f
    a, b = torch.load(os.path.join(out_a, "edited_latents.pt")).float(), torch.load(os.path.join(out_b, "edited_latents.pt")).float()
    assert (a - b).abs().max() <= 5e-2 * a.abs().max()
    assert sorted(os.listdir(out_a)) == sorted(os.listdir(out_b))


def _shard_worker(rank, world, port, base):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ops_emulation as emu
    emu.install()
    os.environ["ANYV2V_NO_GRAPH"] = "1"
    torch.set_grad_enabled(False)
    from anyv2v_amd import run_group_ddim_inversion as s1, run_group_pnp_edit as s2
    inv, inv_list, ed, ed_list = _configs(base, "shard")
    inv_list = [dict(inv_list[0], recon_config={"enable_recon": False})]
    ed_list = [dict(ed_list[0], editing_prompt=f"edit number {i}", edited_video_name=f"edit{i}", pnp_f_t=[0.0, 0.25, 0.5][i % 3],
                    active=(i != 5)) for i in range(9)]       # 8 active edit entries (+ 1 inactive) on 2 ranks: 4 each
    log = logging.getLogger("e2e")
    s1.main(inv, inv_list, torch.device("cpu"), log, synthetic_encoders=True)   # rank 0 inverts the one clip, rank 1 has no entry
    import torch.distributed as dist
    dist.barrier()
    s2.main(ed, ed_list, torch.device("cpu"), log, synthetic_encoders=True)
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sharded_runners_world2_gather_every_entry(tmp_path):
    """BASELINE config 4 in miniature: 8 edit entries dealt round-robin to 2 gloo ranks -- every entry's
    edited latents reach rank 0 through the ONE all_gather, in entry order (round 1 gathered each rank's last entry only)."""
    import torch.multiprocessing as mp
    base = _make_workspace(tmp_path)
    mp.spawn(_shard_worker, args=(2, _free_port(), base), nprocs=2, join=True)
    got = torch.load(os.path.join(base, "gathered_latents.pt"))
    assert tuple(got.shape) == (8, 4, N_FRAMES, SIZE // 8, SIZE // 8)
    names = [f"edit{i}" for i in range(9) if i != 5]
    per_entry = []
    for k, name in enumerate(names):
        root = os.path.join(base, "Results", "Prompt-Based-Editing", "mini-shard", "clip", name)
        sub = os.listdir(root)
        assert len(sub) == 1
        lat = torch.load(os.path.join(root, sub[0], "edited_latents.pt"))
        assert torch.equal(got[k], lat[0]), name
        per_entry.append(lat)
    # the entries really are different jobs (different prompt / schedule / per-entry seed)
    assert all(not torch.equal(per_entry[0], x) for x in per_entry[1:])


def test_fused_runner_matches_the_two_stage_run(tmp_path):
    """``run_group_anyv2v``: stage 1 -> stage 2 in one process (one pipeline, trajectory handed over in HBM)
    writes the same files with the same contents as the two separate stages, and never reads the ``ddim_latents_{t}.pt`` files
    back (they are removed before stage 2 could: the hand-off is the in-memory trajectory)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ops_emulation as emu
    emu.install()
    os.environ["ANYV2V_NO_GRAPH"] = "1"
    torch.set_grad_enabled(False)
    base = _make_workspace(tmp_path)
    _run_both_stages(base, "two")
    from anyv2v_amd import run_group_anyv2v as fused, run_group_pnp_edit as s2
    from anyv2v_amd.utils import LatentTrajectory
    inv, inv_list, ed, ed_list = _configs(base, "fused")
    loads = []
    orig = LatentTrajectory.load
    LatentTrajectory.load = classmethod(lambda cls, *a, **k: loads.append(a) or orig(*a, **k))
    try:
        trajs = fused.main(inv, inv_list, ed, ed_list, torch.device("cpu"), logging.getLogger("e2e"), synthetic_encoders=True)
    finally:
        LatentTrajectory.load = orig
    assert len(trajs) == 1 and not loads          # the edit took the trajectory from memory
    (inv_a, out_a), (inv_b, out_b) = _outputs(base, "two"), _outputs(base, "fused")
    fa, fb = sorted(os.listdir(os.path.join(inv_a, "ddim_latents"))), sorted(os.listdir(os.path.join(inv_b, "ddim_latents")))
    assert fa == fb and len(fa) == N_STEPS
    for f in fa:
        assert torch.equal(torch.load(os.path.join(inv_a, "ddim_latents", f)), torch.load(os.path.join(inv_b, "ddim_latents", f))), f
    assert torch.equal(torch.load(os.path.join(out_a, "edited_latents.pt")), torch.load(os.path.join(out_b, "edited_latents.pt")))
    assert sorted(os.listdir(out_a)) == sorted(os.listdir(out_b))


def _fused_shard_worker(rank, world, port, base):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ops_emulation as emu
    emu.install()
    os.environ["ANYV2V_NO_GRAPH"] = "1"
    torch.set_grad_enabled(False)
    from anyv2v_amd import run_group_anyv2v as fused
    inv, inv_list, ed, ed_list = _configs(base, "fshard")
    inv_list = [dict(inv_list[0], recon_config={"enable_recon": False})]
    ed_list = [dict(ed_list[0], editing_prompt=f"edit number {i}", edited_video_name=f"edit{i}") for i in range(2)]
    fused.main(inv, inv_list, ed, ed_list, torch.device("cpu"), logging.getLogger("e2e"), synthetic_encoders=True)
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_fused_runner_sharded_world2_one_inversion_two_edits(tmp_path):
    """the fused runner under torchrun deals each stage's entries independently -- with 1 clip and 2 edits
    on 2 ranks, rank 1 has no inversion work and its edit needs the files rank 0 is still writing.  The runner now joins the
    writers and barriers between the stages; both edits complete and equal what the same entries give through the files."""
    import torch.multiprocessing as mp
    base = _make_workspace(tmp_path)
    mp.spawn(_fused_shard_worker, args=(2, _free_port(), base), nprocs=2, join=True)
    got = torch.load(os.path.join(base, "gathered_latents.pt"))
    assert tuple(got.shape) == (2, 4, N_FRAMES, SIZE // 8, SIZE // 8)
    lats = []
    for k in range(2):
        root = os.path.join(base, "Results", "Prompt-Based-Editing", "mini-fshard", "clip", f"edit{k}")
        sub = os.listdir(root)
        assert len(sub) == 1
        lat = torch.load(os.path.join(root, sub[0], "edited_latents.pt"))
        assert torch.equal(got[k], lat[0])
        lats.append(lat)
    assert not torch.equal(lats[0], lats[1])
    inv_dir = os.path.join(base, "inversions", "mini-fshard", "clip", "ddim_latents")
    assert len(os.listdir(inv_dir)) == N_STEPS and not [f for f in os.listdir(os.path.dirname(inv_dir)) if "partial" in f]


def _gather_disagree_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from anyv2v_amd.parallel import gather_latents
    dist.init_process_group("gloo")
    shape = (4, 2, 8, 8) if rank == 0 else (4, 2, 4, 4)      # rank 1's clip has another geometry
    try:
        gather_latents([torch.zeros((1,) + shape)], 2, (4, 2, 8, 8), torch.float32, "cpu")
        msg = "no error"
    except ValueError as e:
        msg = "ValueError: " + str(e)[:60]
    open(os.path.join(out, f"rank{rank}.txt"), "w").write(msg)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gather_latents_raises_on_every_rank_together(tmp_path):
    """a geometry mismatch on one rank used to raise there and leave the others in the collective until the
    backend timeout; shapes are agreed on first (all_gather_object), so every rank raises the same error at once."""
    import torch.multiprocessing as mp
    mp.spawn(_gather_disagree_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    msgs = [open(os.path.join(str(tmp_path), f"rank{r}.txt")).read() for r in range(2)]
    assert all(m.startswith("ValueError: gather_latents") for m in msgs), msgs


# ------------------------------------------------------------------------------------------------- pipelined fused runner
def _two_clip_job(base, tag, grouped=False):
    """Two clips (the second: the first one's frames mirrored), three edits in an order that is NOT grouped by clip -- or, with
    ``grouped``, the usual edit list grouped by clip ([clip, clip, clip2]: clip-wise dealing then differs from round-robin dealing)."""
    clip, clip2 = os.path.join(base, "demo", "clip"), os.path.join(base, "demo", "clip2")
    if not os.path.isdir(clip2):
        os.makedirs(os.path.join(clip2, "edited_first_frame"))
        for i in range(N_FRAMES):
            Image.open(os.path.join(clip, f"{i:05d}.png")).transpose(Image.FLIP_LEFT_RIGHT).save(os.path.join(clip2, f"{i:05d}.png"))
        Image.open(os.path.join(clip, "edited_first_frame", "e.png")).transpose(Image.FLIP_TOP_BOTTOM).save(
            os.path.join(clip2, "edited_first_frame", "e.png"))
    inv, inv_list, ed, ed_list = _configs(base, tag)
    inv_list = [dict(inv_list[0]), dict(inv_list[0], video_name="clip2")]
    e = ed_list[0]
    ed_list = [dict(e, edited_video_name="a"), dict(e, video_name="clip2", edited_first_frame_path="demo/clip2/edited_first_frame/e.png",
                                                      edited_video_name="b", editing_prompt="a cat"),
               dict(e, edited_video_name="c", editing_prompt="a dog", pnp_f_t=0.5)]
    if grouped:
        ed_list = [ed_list[0], ed_list[2], ed_list[1]]
    return inv, inv_list, ed, ed_list


def _tree(root):
    out = {}
    for d, _, files in os.walk(root):
        for f in files:
            out[os.path.relpath(os.path.join(d, f), root)] = os.path.join(d, f)
    return out


def _check_pipelined_equals_serial(tmp_path, device, native_vae=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if device == "cpu":
        import cpu_ops_emulation as emu
        emu.install()
        os.environ["ANYV2V_NO_GRAPH"] = "1"
    torch.set_grad_enabled(False)
    base = _make_workspace(tmp_path)
    from anyv2v_amd import run_group_anyv2v as fused
    if native_vae:   # the real AutoencoderKL architecture (full width: 8 x downscale; random weights) on the kernels instead of the stand-in:
        # the pipelined order then runs VAE encodes (next clip) and decodes (current clip) on two streams at once
        from anyv2v_amd import encoders
        synth = encoders.attach_synthetic_encoders

        def attach(pipe):
            synth(pipe)
            encoders.attach_native_vae(pipe, random_init_seed=0)
        fused.attach_synthetic_encoders = attach
    log = logging.getLogger("e2e")
    for tag, pipelined in (("ser", False), ("pipe", True)):
        inv, inv_list, ed, ed_list = _two_clip_job(base, tag)
        inv.device = ed.device = device
        trajs = fused.main(inv, inv_list, ed, ed_list, torch.device(device), log, synthetic_encoders=True, pipelined=pipelined)
        assert len(trajs) == 2
    import filecmp
    for top in ("inversions", os.path.join("Results", "Prompt-Based-Editing")):
        a, b = _tree(os.path.join(base, top, "mini-ser")), _tree(os.path.join(base, top, "mini-pipe"))
        assert sorted(a) == sorted(b) and len(a) > 0
        for rel in a:
            if rel.endswith(".pt"):
                assert torch.equal(torch.load(a[rel]), torch.load(b[rel])), rel
            else:
                assert filecmp.cmp(a[rel], b[rel], shallow=False), rel
    res = _tree(os.path.join(base, "Results", "Prompt-Based-Editing", "mini-pipe"))
    assert len([r for r in res if r.endswith("edited_latents.pt")]) == 3


def test_pipelined_fused_runner_writes_what_the_serial_one_writes(tmp_path, monkeypatch):
    """``run_group_anyv2v`` on one GPU inverts clip k + 1 while clip k is edited (two streams, two pipeline objects around one set of
    weights): every file of a two-clip, three-edit job -- trajectories, reconstructions, edited latents, png / gif / mp4 -- equals
    the serial order's, also for an edit list that is not grouped by clip.  (The pipelined order runs both edits of clip 1 back to
    back, so the second one replays source features from the multi-edit cache where the serial order recomputes them; that is
    bit-equal on the kernels -- the GPU variant below keeps the cache on -- but not on the CPU op emulation, whose matmuls are not
    invariant to the number of rows: the cache is off here.)"""
    monkeypatch.setenv("ANYV2V_SOURCE_CACHE", "0")
    _check_pipelined_equals_serial(tmp_path, "cpu")


@pytest.mark.gpu
def test_pipelined_fused_runner_on_gpu(tmp_path):
    """The same on cuda:0: HIP graphs replayed on two streams side by side, bit-equal files."""
    _check_pipelined_equals_serial(tmp_path, "cuda")


@pytest.mark.gpu
def test_pipelined_fused_runner_on_gpu_with_the_native_vae(tmp_path):
    """... and with the AutoencoderKL on the kernels: encodes of clip k + 1 and decodes of clip k overlap (per-stream GroupNorm scratch)."""
    from anyv2v_amd import run_group_anyv2v as fused
    saved = fused.attach_synthetic_encoders
    try:
        _check_pipelined_equals_serial(tmp_path, "cuda", native_vae=True)
    finally:
        fused.attach_synthetic_encoders = saved


def _clip_shard_worker(rank, world, port, base, tag, pipelined):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ["ANYV2V_SOURCE_CACHE"] = "0"   # (see test_pipelined_fused_runner_writes_what_the_serial_one_writes)
    torch.set_num_threads(2)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ops_emulation as emu
    emu.install()
    os.environ["ANYV2V_NO_GRAPH"] = "1"
    torch.set_grad_enabled(False)
    from anyv2v_amd import run_group_anyv2v as fused
    # edit list grouped by clip: rank 0 holds entries 0 and 1, rank 1 entry 2 -- not the round-robin pattern
    inv, inv_list, ed, ed_list = _two_clip_job(base, tag, grouped=True)
    fused.main(inv, inv_list, ed, ed_list, torch.device("cpu"), logging.getLogger("e2e"), synthetic_encoders=True, pipelined=pipelined)
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.timeout(1200)
def test_fused_runner_world2_deals_whole_clips_and_pipelines_them(tmp_path):
    """Under torchrun with at least as many clips as ranks the fused runner deals whole clips (an inversion + its edits) to the ranks and
    every rank pipelines its own; per-entry seeding makes the files equal to the stage-wise dealing's, and rank 0 gathers every
    entry's latents in entry order either way."""
    import filecmp
    import torch.multiprocessing as mp
    base = _make_workspace(tmp_path)
    for tag, pipelined in (("w2s", False), ("w2p", True)):
        mp.spawn(_clip_shard_worker, args=(2, _free_port(), base, tag, pipelined), nprocs=2, join=True)
        os.replace(os.path.join(base, "gathered_latents.pt"), os.path.join(base, f"gathered_{tag}.pt"))
    a, b = torch.load(os.path.join(base, "gathered_w2s.pt")), torch.load(os.path.join(base, "gathered_w2p.pt"))
    assert tuple(a.shape) == (3, 4, N_FRAMES, SIZE // 8, SIZE // 8) and torch.equal(a, b)
    # gathered_latents.pt is in ENTRY order under both dealings: entry k of the file is the edited_latents.pt of edit k
    for k, name in enumerate(("a", "c", "b")):
        roots = [r for r in _tree(os.path.join(base, "Results", "Prompt-Based-Editing", "mini-w2p")) if r.endswith("edited_latents.pt") and
                 r.split(os.sep)[1] == name]
        assert len(roots) == 1, roots
        lat = torch.load(os.path.join(base, "Results", "Prompt-Based-Editing", "mini-w2p", roots[0]))
        assert torch.equal(b[k], lat[0]), (k, name)
    for top in ("inversions", os.path.join("Results", "Prompt-Based-Editing")):
        x, y = _tree(os.path.join(base, top, "mini-w2s")), _tree(os.path.join(base, top, "mini-w2p"))
        assert sorted(x) == sorted(y) and len(x) > 0
        for rel in x:
            if rel.endswith(".pt"):
                assert torch.equal(torch.load(x[rel]), torch.load(y[rel])), rel
            else:
                assert filecmp.cmp(x[rel], y[rel], shallow=False), rel


def _check_batched_inversion_job(tmp_path, device):
    """The two-clip, three-edit job with ``batch_clips=2`` (both inversions in one B = 2 forward per step) against the one-by-one serial
    run: same files, trajectories and edited latents equal up to the rounding of another launch plan."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if device == "cpu":
        import cpu_ops_emulation as emu
        emu.install()
        os.environ["ANYV2V_NO_GRAPH"] = "1"
    torch.set_grad_enabled(False)
    base = _make_workspace(tmp_path)
    from anyv2v_amd import run_group_anyv2v as fused
    log = logging.getLogger("e2e")
    for tag, bc in (("one", 1), ("two", 2)):
        inv, inv_list, ed, ed_list = _two_clip_job(base, tag)
        inv.device = ed.device = device
        trajs = fused.main(inv, inv_list, ed, ed_list, torch.device(device), log, synthetic_encoders=True, pipelined=False, batch_clips=bc)
        assert len(trajs) == 2
    for top in ("inversions", os.path.join("Results", "Prompt-Based-Editing")):
        a, b = _tree(os.path.join(base, top, "mini-one")), _tree(os.path.join(base, top, "mini-two"))
        assert sorted(a) == sorted(b) and len(a) > 0
        for rel in a:
            if rel.endswith(".pt"):
                x, y = torch.load(a[rel]).float(), torch.load(b[rel]).float()
                assert float((x - y).abs().max()) <= 3e-2 * float(x.abs().max()), rel


def test_batched_inversion_job_matches_the_one_by_one_run(tmp_path, monkeypatch):
    monkeypatch.setenv("ANYV2V_SOURCE_CACHE", "0")
    _check_batched_inversion_job(tmp_path, "cpu")


@pytest.mark.gpu
def test_batched_inversion_job_on_gpu(tmp_path):
    _check_batched_inversion_job(tmp_path, "cuda")
