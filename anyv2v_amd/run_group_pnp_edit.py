"""Stage 2 CLI -- PnP edit of a list of (clip, edited first frame) entries (same flags / config keys / output files as
the reference's ``i2vgen-xl/run_group_pnp_edit.py``):

    python -m anyv2v_amd.run_group_pnp_edit --template_config configs/group_pnp_edit/template.yaml \
                                            --configs_json configs/group_pnp_edit/group_config.json

Under ``torchrun --nproc-per-node N`` entries are dealt round-robin to the ranks; after the local loop the edited
latents of every rank's last entry are exchanged with ONE all_gather (RCCL over xGMI) and rank 0
writes ``gathered_latents.pt`` next to the outputs.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
from pathlib import Path

import torch
from PIL import Image

from .config import OmegaConf
from .encoders import attach_synthetic_encoders
from .parallel import FrameParallel, gather_latents, init_distributed, seed_for_entry, shard_entries
from .pipeline import I2VGenXLPipeline
from .pnp_utils import register_conv_injection, register_spatial_attention_pnp, register_temp_attention_pnp
from .schedulers import DDIMScheduler
from .utils import (LatentTrajectory, convert_video_to_frames, export_to_gif, export_to_video, load_ddim_latents_at_t, load_image,
                    load_video_frames, seed_everything)

MODEL_ID = "ali-vilab/i2vgen-xl"


def init_pnp(pipe, scheduler, config):
    """``run_group_pnp_edit.py:35-56``: int() truncation; schedules are prefixes of the FULL timestep list."""
    conv_injection_t = int(config.n_steps * config.pnp_f_t)
    spatial_attn_qk_injection_t = int(config.n_steps * config.pnp_spatial_attn_t)
    temp_attn_qk_injection_t = int(config.n_steps * config.pnp_temp_attn_t)
    ts = scheduler.timesteps
    register_conv_injection(pipe, ts[:conv_injection_t] if conv_injection_t >= 0 else [])
    register_spatial_attention_pnp(pipe, ts[:spatial_attn_qk_injection_t] if spatial_attn_qk_injection_t >= 0 else [])
    register_temp_attention_pnp(pipe, ts[:temp_attn_qk_injection_t] if temp_attn_qk_injection_t >= 0 else [])
    logger = logging.getLogger(__name__)
    logger.debug(f"conv_injection_t: {conv_injection_t}")
    logger.debug(f"spatial_attn_qk_injection_t: {spatial_attn_qk_injection_t}")
    logger.debug(f"temp_attn_qk_injection_t: {temp_attn_qk_injection_t}")


def output_suffix(config, ddim_init_latents_t_idx) -> str:
    """``run_group_pnp_edit.py:154-167``."""
    return ("ddim_init_latents_t_idx_" + str(ddim_init_latents_t_idx) + "_nsteps_" + str(config.n_steps) + "_cfg_"
            + str(config.cfg) + "_pnpf" + str(config.pnp_f_t) + "_pnps" + str(config.pnp_spatial_attn_t) + "_pnpt"
            + str(config.pnp_temp_attn_t))


class Stage2:
    """Stage 2 over a list of entries: ``my_entries`` (this rank's share, list order), ``run_entry`` (one edit, files written),
    ``latents_dir_of`` (which inversion an entry edits: ``run_group_anyv2v`` schedules the edits of a clip right behind its
    inversion), ``finish`` (the cross-rank gather).  ``main`` runs them in list order."""

    def __init__(self, template_config, configs_list, device, logger, synthetic_encoders=False, random_init_seed=None,
                 frame_parallel=False, pipe=None, trajectories=None, my_entries=None):
        self.template_config, self.configs_list, self.device, self.logger = template_config, configs_list, device, logger
        self._my_entries = my_entries     # this rank's share when the caller deals the entries itself (clip-wise dealing)
        self.trajectories = trajectories
        self.rank, self.local_rank, self.world = init_distributed()
        # --frame_parallel (long clips): every rank works on EVERY entry, the clip's frames sharded over
        # the ranks inside the UNet (parallel.FrameParallel); inputs, latents and RNG draws are replicated; rank 0 writes.
        self.fp_mode = bool(frame_parallel) and self.world > 1
        self.writer = self.rank == 0 or not self.fp_mode
        self.e_rank, self.e_world = (0, 1) if self.fp_mode else (self.rank, self.world)
        if pipe is None:
            pipe = I2VGenXLPipeline.from_pretrained(template_config.get("model_path", MODEL_ID), torch_dtype=torch.float16,
                                                    variant="fp16", random_init_seed=random_init_seed)
            pipe.to(device)
            if synthetic_encoders:
                attach_synthetic_encoders(pipe)
            if self.fp_mode:
                pipe.unet.set_frame_parallel(FrameParallel())
        self.pipe = pipe
        self.ddim_scheduler = DDIMScheduler.from_pretrained(MODEL_ID, subfolder="scheduler")
        self.all_active = [e for e in configs_list if e["active"] is not False]
        for config_entry in configs_list:
            if config_entry["active"] is False:
                logger.info(f"Skipping config_entry: {config_entry}")
        self.my_latents, self.lat_shape = {}, None
        self.my_entries = self._my_entries if self._my_entries is not None else shard_entries(configs_list, self.e_rank, self.e_world)
        # Several edits of one clip on this rank (the demo group config: 8 edits of one clip): keep the source branch's injected features
        # in HBM after the first edit, so that the further ones run [negative, editing] only (pipeline.SourceFeatureCache; exact).
        # ANYV2V_SOURCE_CACHE=0 switches it off.
        clips = [e.get("video_name") for e in self.my_entries]
        if os.environ.get("ANYV2V_SOURCE_CACHE", "1") == "1" and len(clips) != len(set(clips)) and pipe.source_cache is None:
            pipe.enable_source_cache(True)
        self.loaded_trajectories = {}   # one LatentTrajectory object per clip (read once, shared by the clip's edits)

    def _config(self, config_entry):
        config = OmegaConf.merge(self.template_config, OmegaConf.create(config_entry))
        config.video_path = os.path.join(config.video_dir, config.video_name + ".mp4")
        config.video_frames_path = os.path.join(config.video_dir, config.video_name)
        config.edited_first_frame_path = os.path.join(config.data_dir, config.edited_first_frame_path)
        return config

    def latents_dir_of(self, config_entry) -> str:
        return os.path.abspath(str(self._config(config_entry).ddim_latents_path))

    def run_entry(self, config_entry):
        for _ in self.entry(config_entry):
            pass

    def entry(self, config_entry):
        """Generator: runs the entry up to the point where the 50 edit steps are ENQUEUED (nothing read back) and yields; then
        decodes and writes the files."""
        from .run_group_ddim_inversion import _RngState
        template_config, logger, pipe, device = self.template_config, self.logger, self.pipe, self.device
        ddim_scheduler, loaded_trajectories = self.ddim_scheduler, self.loaded_trajectories
        entry_idx = self.all_active.index(config_entry)
        logger.info(f"[rank {self.rank}/{self.world}] Processing config_entry: {config_entry}")
        config = self._config(config_entry)
        logger.info(f"config: {OmegaConf.to_yaml(config)}")
        for k, v in config.items():  # logs only -- the reference's `continue` continues this inner loop (:89-93)
            if "ReplaceMe" in str(v):
                logger.error(f"Field {k} contains 'ReplaceMe'")
        try:
            logger.info(f"Loading frames from: {config.video_frames_path}")
            _, frame_list = load_video_frames(config.video_frames_path, config.n_frames, tuple(config.image_size))
        except Exception:
            logger.error(f"Failed to load frames from: {config.video_frames_path}")
            frame_list = convert_video_to_frames(config.video_path, tuple(config.image_size), save_frames=True)
            frame_list = frame_list[: config.n_frames]
        src_1st_frame = frame_list[0]
        edited_1st_frame = load_image(config.edited_first_frame_path)
        edited_1st_frame = edited_1st_frame.resize(tuple(config.image_size), resample=Image.Resampling.LANCZOS)

        t_idx = config.ddim_init_latents_t_idx
        ddim_scheduler.set_timesteps(config.n_steps)
        logger.info(f"ddim_scheduler.timesteps: {ddim_scheduler.timesteps}")
        # read the whole source trajectory once into HBM (the reference re-reads one file per step, :1134)
        tkey = (os.path.abspath(str(config.ddim_latents_path)), int(config.n_steps), int(t_idx))
        handed = (self.trajectories or {}).get(tkey[0])
        if handed is not None:
            traj = handed
        else:
            if tkey not in loaded_trajectories:
                loaded_trajectories.clear()   # (one clip's trajectory at a time: 50 x 512 KiB at 16 f x 512^2)
                loaded_trajectories[tkey] = LatentTrajectory.load(
                    config.ddim_latents_path, device=device, timesteps=[int(t) for t in ddim_scheduler.timesteps[t_idx:]])
            traj = loaded_trajectories[tkey]
        ddim_latents_at_t = load_ddim_latents_at_t(ddim_scheduler.timesteps[t_idx], traj)
        seed_everything(seed_for_entry(template_config.seed, entry_idx) if self.e_world > 1 else template_config.seed)
        random_latents = torch.randn(ddim_latents_at_t.shape, dtype=torch.float32).to(ddim_latents_at_t)  # drawn even if unused (:124)
        logger.info(f"Blending random_ratio (1 means random latent): {config.random_ratio}")
        mixed_latents = random_latents * config.random_ratio + ddim_latents_at_t * (1 - config.random_ratio)

        init_pnp(pipe, ddim_scheduler, config)
        pipe.register_modules(scheduler=ddim_scheduler)
        edited_latents = pipe.sample_with_pnp(
            prompt=config.editing_prompt, image=edited_1st_frame, height=config.image_size[1], width=config.image_size[0],
            num_frames=config.n_frames, num_inference_steps=config.n_steps, guidance_scale=config.cfg,
            negative_prompt=config.editing_negative_prompt, target_fps=config.target_fps, latents=mixed_latents,
            generator=torch.manual_seed(config.seed), return_dict=True, ddim_init_latents_t_idx=t_idx,
            ddim_inv_latents_path=traj, ddim_inv_prompt=config.ddim_inv_prompt, ddim_inv_1st_frame=src_1st_frame,
            output_type="latent").frames
        self.my_latents[entry_idx] = edited_latents
        self.lat_shape = tuple(edited_latents.shape)
        rng = _RngState(device)
        yield
        rng.restore()
        video = pipe.decode_latents(edited_latents, decode_chunk_size=1)  # (frame-parallel: every rank decodes its frames)
        if not self.writer:
            return
        edited_video = pipe.vae.to_pil(video)

        output_dir = os.path.join(config.output_dir, output_suffix(config, t_idx))
        os.makedirs(output_dir, exist_ok=True)
        edited_video = [frame.resize(tuple(config.image_size), resample=Image.LANCZOS) for frame in edited_video]
        name = "video"
        export_to_video(edited_video, os.path.join(output_dir, f"{name}.mp4"), fps=config.target_fps)
        export_to_gif(edited_video, os.path.join(output_dir, f"{name}.gif"))      # (no rate, as ``:179``: 100 ms per frame)
        logger.info(f"Saved video to: {os.path.join(output_dir, f'{name}.mp4')}")
        logger.info(f"Saved gif to: {os.path.join(output_dir, f'{name}.gif')}")
        for i, frame in enumerate(edited_video):
            frame.save(os.path.join(output_dir, f"{name}_{i:05d}.png"))
        torch.save(edited_latents.cpu(), os.path.join(output_dir, "edited_latents.pt"))

    def finish(self):
        template_config, device = self.template_config, self.device
        my_idx = sorted(self.my_latents)                                     # entry order, whatever order they were run in
        my_latents = [self.my_latents[k] for k in my_idx]
        if self.fp_mode:
            import torch.distributed as dist
            dist.barrier()
        elif self.world > 1:
            import torch.distributed as dist
            shape = self.lat_shape or (1, 4, template_config.n_frames, template_config.image_size[1] // 8, template_config.image_size[0] // 8)
            # every entry's latents reach rank 0, in entry order (a rank may have run several entries, or none)
            # (the entry indices travel with the latents: clip-wise dealing -- run_group_anyv2v under torchrun -- is not round-robin)
            gathered = gather_latents(my_latents, len(self.all_active), shape, torch.float16, device, indices=my_idx)
            if self.rank == 0:
                out = os.path.join(template_config.get("data_dir", "."), "gathered_latents.pt")
                torch.save(gathered.cpu(), out)
                self.logger.info(f"all_gather of the edited latents of {len(self.all_active)} entries from {self.world} ranks -> {out}")
            dist.barrier()


def main(template_config, configs_list, device, logger, synthetic_encoders=False, random_init_seed=None, frame_parallel=False,
         pipe=None, trajectories=None):
    """``pipe`` / ``trajectories``: see ``run_group_ddim_inversion.main`` -- an entry whose ``ddim_latents_path`` is a key of
    ``trajectories`` takes the source trajectory from HBM instead of reading the ``ddim_latents_{t}.pt`` files."""
    stage = Stage2(template_config, configs_list, device, logger, synthetic_encoders, random_init_seed, frame_parallel, pipe, trajectories)
    for config_entry in stage.my_entries:
        stage.run_entry(config_entry)
    stage.finish()


def cli(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--template_config", type=str, default="./configs/group_pnp_edit/template.yaml")
    parser.add_argument("--configs_json", type=str, default="./configs/group_config.json")
    parser.add_argument("--synthetic_encoders", action="store_true")
    parser.add_argument("--random_init_seed", type=int, default=None)
    parser.add_argument("--frame_parallel", action="store_true",
                        help="under torchrun: shard every clip's frames over the ranks instead of dealing entries to ranks")
    args = parser.parse_args(argv)
    template_config = OmegaConf.load(args.template_config)
    logging_level = logging.DEBUG if template_config.debug else logging.INFO
    logging.basicConfig(level=logging_level, format="%(asctime)s - %(levelname)s - [%(funcName)s] - %(message)s")
    logger = logging.getLogger(__name__)
    logger.info(f"template_config: {OmegaConf.to_yaml(template_config)}")
    assert Path(args.configs_json).exists()
    with open(args.configs_json, "r") as f:
        configs_list = json.load(f)
    logger.info(f"Loaded {len(configs_list)} configs from {args.configs_json}")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = template_config.device if world == 1 else f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
    device = torch.device(dev)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    torch.set_grad_enabled(False)
    seed_everything(template_config.seed)
    main(template_config, configs_list, device, logger, args.synthetic_encoders, args.random_init_seed, args.frame_parallel)


if __name__ == "__main__":
    cli()
