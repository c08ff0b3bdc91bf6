"""Stage 1 CLI -- DDIM inversion of a list of clips (same flags / config keys / output files as the reference's
``i2vgen-xl/run_group_ddim_inversion.py``):

    python -m anyv2v_amd.run_group_ddim_inversion --template_config configs/group_ddim_inversion/template.yaml \
                                                  --configs_json configs/group_ddim_inversion/group_config.json

Under ``torchrun --nproc-per-node N`` the active entries are dealt round-robin to the ranks (one clip per GPU at a
time); each rank writes its own ``ddim_latents_{t}.pt`` files, so stage 1 needs no collective.
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
from .parallel import FrameParallel, init_distributed, seed_for_entry, shard_entries
from .pipeline import I2VGenXLPipeline
from .schedulers import DDIMInverseScheduler, DDIMScheduler
from .utils import (convert_video_to_frames, export_to_gif, export_to_video, inversion_is_complete, load_ddim_latents_at_t, load_video_frames,
                    seed_everything)

MODEL_ID = "ali-vilab/i2vgen-xl"


def ddim_inversion(config, first_frame, frame_list, pipe: I2VGenXLPipeline, inverse_scheduler, g, write=True):
    """``run_group_ddim_inversion.py:29-55``."""
    pipe.scheduler = inverse_scheduler
    video_latents_at_0 = pipe.encode_vae_video(frame_list, device=pipe._execution_device, height=config.image_size[1],
                                               width=config.image_size[0])
    ddim_latents = pipe.invert(prompt=config.prompt, image=first_frame, height=config.image_size[1],
                               width=config.image_size[0], num_frames=config.n_frames,
                               num_inference_steps=config.n_steps, guidance_scale=config.cfg,
                               negative_prompt=config.negative_prompt, target_fps=config.target_fps,
                               latents=video_latents_at_0, generator=g, return_dict=False,
                               output_dir=config.output_dir if write else None, background_save=True)
    logging.getLogger(__name__).debug(f"ddim_latents.shape: {ddim_latents.shape}")
    return ddim_latents[0]  # [num_inference_steps, c, num_frames, h, w]


def ddim_sampling(config, first_frame, ddim_latents_at_T, pipe: I2VGenXLPipeline, ddim_scheduler, ddim_init_latents_t_idx, g):
    """``run_group_ddim_inversion.py:58-77``."""
    pipe.scheduler = ddim_scheduler
    return pipe(prompt=config.prompt, image=first_frame, height=config.image_size[1], width=config.image_size[0],
                num_frames=config.n_frames, num_inference_steps=config.n_steps, guidance_scale=config.cfg,
                negative_prompt=config.negative_prompt, target_fps=config.target_fps, latents=ddim_latents_at_T,
                generator=g, return_dict=True, ddim_init_latents_t_idx=ddim_init_latents_t_idx).frames[0]


class _RngState:
    """Host + device RNG state of this process (the pipelined runner parks an inversion entry between its enqueue and its finish
    while an edit entry re-seeds everything)."""

    def __init__(self, device):
        import random

        import numpy as np
        self.device = device
        self.py, self.np, self.cpu = random.getstate(), np.random.get_state(), torch.get_rng_state()
        self.cuda = torch.cuda.get_rng_state(device) if device.type == "cuda" else None

    def restore(self):
        import random

        import numpy as np
        random.setstate(self.py)
        np.random.set_state(self.np)
        torch.set_rng_state(self.cpu)
        if self.cuda is not None:
            torch.cuda.set_rng_state(self.cuda, self.device)


class Stage1:
    """Stage 1 over a list of entries.  ``entries()`` yields one generator per entry this rank works on; a generator runs the
    entry up to the point where the inversion is ENQUEUED (frames loaded, VAE encode, 50 steps launched, nothing read back) and
    yields the absolute latents directory -- ``run_group_anyv2v`` edits the previous clip at that point --, then finishes it
    (files, reconstruction).  ``main`` simply exhausts them one after the other."""

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
        self.inverse_scheduler = DDIMInverseScheduler.from_pretrained(MODEL_ID, subfolder="scheduler")
        self.ddim_scheduler = DDIMScheduler.from_pretrained(MODEL_ID, subfolder="scheduler")
        video_dir = template_config.video_dir
        assert os.path.exists(video_dir), f"video_dir: {video_dir} does not exist"
        self.all_active = [e for e in configs_list if e["active"] is not False]
        for config_entry in configs_list:
            if config_entry["active"] is False:
                logger.info(f"Skipping config_entry: {config_entry}")

    def entries(self):
        mine = self._my_entries if self._my_entries is not None else shard_entries(self.configs_list, self.e_rank, self.e_world)
        for config_entry in mine:
            yield self._entry(config_entry)

    def _prepare(self, config_entry):
        """Config, frames and seeds of one entry; None when the entry is skipped (complete on disk)."""
        template_config, logger = self.template_config, self.logger
        rank, world, fp_mode = self.rank, self.world, self.fp_mode
        entry_idx = self.all_active.index(config_entry)
        logger.info(f"[rank {rank}/{world}] Processing config_entry: {config_entry}")
        config = OmegaConf.merge(template_config, OmegaConf.create(config_entry))
        config.video_path = os.path.join(config.video_dir, config.video_name + ".mp4")
        config.video_frames_path = os.path.join(config.video_dir, config.video_name)
        skip = (inversion_is_complete(config.output_dir, config.inverse_config.output_dir)
                and not config.get("force_recompute_latents", False))
        if fp_mode:  # all ranks must take the same decision, before rank 0 starts writing into that directory
            import torch.distributed as dist
            flag = [skip]
            dist.broadcast_object_list(flag, src=0)
            skip = flag[0]
        if skip:
            logger.info(f"### Skipping !!! {config.output_dir} already exists. ")
            return None
        logger.info(f"config: {OmegaConf.to_yaml(config)}")
        try:
            logger.info(f"Loading frames from: {config.video_frames_path}")
            _, frame_list = load_video_frames(config.video_frames_path, config.n_frames, tuple(config.image_size))
        except Exception:
            logger.error(f"Failed to load frames from: {config.video_frames_path}")
            logger.info(f"Converting mp4 video to frames: {config.video_path}")
            frame_list = convert_video_to_frames(config.video_path, tuple(config.image_size), save_frames=True)
            frame_list = frame_list[: config.n_frames]
            export_to_gif(frame_list, os.path.join(config.video_frames_path, config.video_name + ".gif"))
        first_frame = frame_list[0]
        if config.inverse_config.inverse_static_video:
            logger.info("### Inverse a static video!")
            frame_list = [frame_list[0]] * config.n_frames
        if config.inverse_config.null_image_inversion:
            logger.info("### Inverse a null image!")
            first_frame = Image.new("RGB", (config.image_size[0], config.image_size[1]), (0, 0, 0))
        seed = seed_for_entry(template_config.seed, entry_idx) if self.e_world > 1 else template_config.seed
        return dict(config=config, first_frame=first_frame, frame_list=frame_list, seed=seed,
                    latents_dir=os.path.abspath(str(config.inverse_config.output_dir)))

    def _finish(self, p, traj, g):
        """Files of one inverted clip (background writer) and its reconstruction."""
        config, logger, pipe, writer, ddim_scheduler = p["config"], self.logger, self.pipe, self.writer, self.ddim_scheduler
        latents_dir, first_frame = p["latents_dir"], p["first_frame"]
        if writer:
            # ddim_latents_{t}.pt, reference format, from a background thread (every reader in anyv2v_amd.utils joins it first)
            traj.save(config.inverse_config.output_dir, background=True)
            logger.info(f"saving noisy latents for {len(traj)} timesteps to {config.inverse_config.output_dir}")
        recon_config = config.recon_config
        if recon_config.enable_recon:
            t_idx = recon_config.ddim_init_latents_t_idx
            ddim_scheduler.set_timesteps(recon_config.n_steps)
            logger.info(f"ddim_scheduler.timesteps: {ddim_scheduler.timesteps}")
            # ``recon_config.ddim_latents_path`` (``run_group_ddim_inversion.py:156-160``): when it names the directory this
            # very inversion writes (the template's default), hand the trajectory over in memory -- the files are still
            # being written in the background; any other directory is read from disk as configured
            src = recon_config.get("ddim_latents_path", None)
            same = src is None or os.path.abspath(str(src)) == latents_dir
            ddim_latents_at_t = load_ddim_latents_at_t(ddim_scheduler.timesteps[t_idx], traj if same else str(src))
            reconstructed_video = ddim_sampling(recon_config, first_frame, ddim_latents_at_t, pipe, ddim_scheduler, t_idx, g)
            if writer:
                traj.wait()  # the latents directory is renamed into place when complete
                os.makedirs(config.output_dir, exist_ok=True)
                reconstructed_video = [f.resize((512, 512), resample=Image.LANCZOS) for f in reconstructed_video]
                export_to_video(reconstructed_video, os.path.join(config.output_dir, "ddim_reconstruction.mp4"), fps=10)   # (:183-187)
                export_to_gif(reconstructed_video, os.path.join(config.output_dir, "ddim_reconstruction.gif"))             # (:188-191)
                logger.info(f"Saved reconstructed video to {config.output_dir}")
        traj.wait()

    def _entry(self, config_entry):
        p = self._prepare(config_entry)
        if p is None:
            return
        pipe = self.pipe
        seed_everything(p["seed"])
        g = torch.Generator().manual_seed(self.template_config.seed)
        # launched, not read back: the files are written below (``write=False`` keeps the trajectory in HBM only for now)
        ddim_inversion(p["config"].inverse_config, p["first_frame"], p["frame_list"], pipe, self.inverse_scheduler, g, write=False)
        traj = pipe._last_trajectory
        if self.trajectories is not None:
            self.trajectories[p["latents_dir"]] = traj
        rng = _RngState(self.device)
        yield p["latents_dir"]
        rng.restore()
        self._finish(p, traj, g)

    def run_batched(self, batch_clips: int):
        """``--batch_clips N``: up to N consecutive entries of the same geometry / step count (guidance 1) are inverted in ONE batch
        (``pipe.invert_clips``: every weight read once per step for all of them).  Per-entry seeding and VAE sampling as in the one-by-one
        path; the trajectories differ from it at rounding level (other launch plans at other row counts)."""
        mine = self._my_entries if self._my_entries is not None else shard_entries(self.configs_list, self.e_rank, self.e_world)
        prepared = [p for p in (self._prepare(e) for e in mine) if p is not None]

        def key(p):
            c = p["config"].inverse_config
            return (tuple(c.image_size), int(c.n_frames), int(c.n_steps), int(c.target_fps), float(c.cfg) == 1.0)
        i = 0
        while i < len(prepared):
            group = [prepared[i]]
            while len(group) < batch_clips and i + len(group) < len(prepared) and key(prepared[i + len(group)]) == key(group[0]) and key(group[0])[-1]:
                group.append(prepared[i + len(group)])
            i += len(group)
            pipe = self.pipe
            g = torch.Generator().manual_seed(self.template_config.seed)
            if len(group) == 1 or not key(group[0])[-1]:
                for p in group:
                    seed_everything(p["seed"])
                    ddim_inversion(p["config"].inverse_config, p["first_frame"], p["frame_list"], pipe, self.inverse_scheduler, g, write=False)
                    trajs = [pipe._last_trajectory]
                    if self.trajectories is not None:
                        self.trajectories[p["latents_dir"]] = trajs[0]
                    self._finish(p, trajs[0], g)
                continue
            c0 = group[0]["config"].inverse_config
            pipe.scheduler = self.inverse_scheduler
            clips = []
            for p in group:
                seed_everything(p["seed"])
                c = p["config"].inverse_config
                lat0 = pipe.encode_vae_video(p["frame_list"], device=pipe._execution_device, height=c.image_size[1], width=c.image_size[0])
                clips.append(dict(prompt=c.prompt, negative_prompt=c.negative_prompt, image=p["first_frame"], latents=lat0))
            self.logger.info(f"inverting {len(group)} clips in one batch: {[p['config'].video_name for p in group]}")
            trajs = pipe.invert_clips(clips, height=c0.image_size[1], width=c0.image_size[0], num_frames=c0.n_frames,
                                      num_inference_steps=c0.n_steps, target_fps=c0.target_fps)
            for p, traj in zip(group, trajs):
                if self.trajectories is not None:
                    self.trajectories[p["latents_dir"]] = traj
                self._finish(p, traj, g)


def main(template_config, configs_list, device, logger, synthetic_encoders=False, random_init_seed=None, frame_parallel=False,
         pipe=None, trajectories=None, batch_clips=1):
    """``pipe``: reuse a pipeline that is already built (``run_group_anyv2v``: both stages in one process); ``trajectories``: dict
    filled with {absolute latents directory: LatentTrajectory} of every inversion run here (the in-HBM hand-off to stage 2);
    ``batch_clips``: see ``Stage1.run_batched``."""
    stage = Stage1(template_config, configs_list, device, logger, synthetic_encoders, random_init_seed, frame_parallel, pipe, trajectories)
    if batch_clips and int(batch_clips) > 1 and not stage.fp_mode:
        return stage.run_batched(int(batch_clips))
    for entry in stage.entries():
        for _ in entry:
            pass


def cli(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--template_config", type=str, default="./configs/group_ddim_inversion/template.yaml")
    parser.add_argument("--configs_json", type=str, default="./configs/group_config.json")
    parser.add_argument("--synthetic_encoders", action="store_true",
                        help="use the weight-free stand-ins for VAE/CLIP (no pretrained weights offline)")
    parser.add_argument("--frame_parallel", action="store_true",
                        help="under torchrun: shard every clip's frames over the ranks instead of dealing clips to ranks")
    parser.add_argument("--random_init_seed", type=int, default=None, help="random UNet weights (no checkpoint offline)")
    parser.add_argument("--batch_clips", type=int, default=1,
                        help="invert up to N consecutive clips of the same geometry in one batch (inversion-bound jobs: the 500-step template)")
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
    main(template_config, configs_list, device, logger, args.synthetic_encoders, args.random_init_seed, args.frame_parallel,
         batch_clips=args.batch_clips)


if __name__ == "__main__":
    cli()
