"""The callable behind the reference's front-ends -- SURVEY.md 8(f) F4 (the Gradio / Cog wrappers, not their UI).

``gradio_demo.py:58-222`` (``AnyV2V_I2VGenXL.perform_anyv2v``) and ``predict.py:43-258`` both do the same thing in one
process: read a short clip, DDIM-invert it, load the edited first frame, blend / pick the start latent, ``init_pnp``,
``sample_with_pnp``, write ``edited_video.mp4``.  This module is that function on the native pipeline, with the same argument
names and meaning, so a front-end imports it instead of assembling the steps itself.  Differences that do not change results:

* the pipeline and the schedulers are built once per object (the reference rebuilds them on every call, ``gradio_demo.py:92-108``);
* the inversion trajectory goes to the edit in HBM (``LatentTrajectory``); the ``ddim_latents_{t}.pt`` files are still written
  under ``<tmp_dir>/ddim_latents`` in the background, same names and format;
* the clip is read with ``anyv2v_amd.mp4`` (this image has no imageio / ffmpeg): an mp4 written by ``export_to_video`` or a
  directory of ``%05d.png`` frames; the result is written by ``export_to_video``.
"""
from __future__ import annotations

import os
import shutil
from typing import List, Optional

import torch
from PIL import Image

from .config import OmegaConf
from .pipeline import I2VGenXLPipeline
from .run_group_ddim_inversion import ddim_inversion
from .run_group_pnp_edit import init_pnp
from .schedulers import DDIMInverseScheduler, DDIMScheduler
from .utils import export_to_video, load_image, wait_for_pending_writes

MODEL_ID = "ali-vilab/i2vgen-xl"


def read_frames(video_path: str) -> List[Image.Image]:
    """``gradio_demo.py:120-127``: every frame of the clip at its own size (no resize)."""
    if os.path.isdir(video_path):
        names = sorted(n for n in os.listdir(video_path) if n.lower().endswith(".png"))
        if not names:
            raise ValueError(f"no .png frames in {video_path}")
        return [load_image(os.path.join(video_path, n)) for n in names]
    from .mp4 import Mp4Unsupported, read_mp4
    try:
        return read_mp4(str(video_path))[0]
    except Mp4Unsupported as e:
        raise RuntimeError(f"cannot decode {video_path}: {e}; no video decoder library is available here -- pass a directory of "
                           "%05d.png frames instead") from e


class AnyV2V_I2VGenXL:
    def __init__(self, model_path: str = MODEL_ID, device="cuda:0", tmp_dir: str = "_demo_temp", pipe: Optional[I2VGenXLPipeline] = None,
                 random_init_seed: Optional[int] = None, synthetic_encoders: bool = False) -> None:
        # default inversion / edit configuration of the demo (gradio_demo.py:60-78)
        self.config = OmegaConf.create({
            "inverse_config": {"image_size": [512, 512], "n_frames": 16, "cfg": 1.0, "target_fps": 8, "ddim_inv_prompt": "",
                               "prompt": "", "negative_prompt": ""},
            "pnp_config": {"random_ratio": 0.0, "target_fps": 8},
        })
        self.device = torch.device(device)
        self.tmp_dir = tmp_dir
        if pipe is None:
            pipe = I2VGenXLPipeline.from_pretrained(model_path, torch_dtype=torch.float16, variant="fp16",
                                                    random_init_seed=random_init_seed)
            pipe.to(self.device)
            if synthetic_encoders:
                from .encoders import attach_synthetic_encoders
                attach_synthetic_encoders(pipe)
        self.pipe = pipe
        self.inverse_scheduler = DDIMInverseScheduler.from_pretrained(MODEL_ID, subfolder="scheduler")
        self.ddim_scheduler = DDIMScheduler.from_pretrained(MODEL_ID, subfolder="scheduler")

    @torch.no_grad()
    def perform_anyv2v(self, video_path, video_prompt, video_negative_prompt, edited_first_frame_path, conv_inj, spatial_inj,
                       temp_inj, num_inference_steps, guidance_scale, ddim_init_latents_t_idx, ddim_inversion_steps, seed):
        """``gradio_demo.py:80-222``.  Returns the path of ``edited_video.mp4``."""
        tmp_dir = os.path.join(self.tmp_dir, "AnyV2V")
        wait_for_pending_writes()  # a previous call that raised may still have a trajectory writer staging files in there
        if os.path.exists(tmp_dir):
            shutil.rmtree(tmp_dir)
        os.makedirs(tmp_dir)
        try:
            return self._perform(tmp_dir, video_path, video_prompt, video_negative_prompt, edited_first_frame_path, conv_inj,
                                 spatial_inj, temp_inj, num_inference_steps, guidance_scale, ddim_init_latents_t_idx,
                                 ddim_inversion_steps, seed)
        finally:
            wait_for_pending_writes()  # also when a later step raised: no writer thread outlives the call

    def _perform(self, tmp_dir, video_path, video_prompt, video_negative_prompt, edited_first_frame_path, conv_inj, spatial_inj,
                 temp_inj, num_inference_steps, guidance_scale, ddim_init_latents_t_idx, ddim_inversion_steps, seed):
        ddim_latents_path = os.path.join(tmp_dir, "ddim_latents")
        frame_list = read_frames(str(video_path))
        cfg = self.config
        cfg.inverse_config.image_size = list(frame_list[0].size)
        cfg.inverse_config.n_steps = ddim_inversion_steps
        cfg.inverse_config.n_frames = len(frame_list)
        cfg.inverse_config.output_dir = ddim_latents_path
        ddim_init_latents_t_idx = min(ddim_init_latents_t_idx, num_inference_steps - 1)

        # Step 1. DDIM inversion
        first_frame = frame_list[0]
        generator = torch.Generator(device=self.device).manual_seed(seed)
        ddim_inversion(cfg.inverse_config, first_frame, frame_list, self.pipe, self.inverse_scheduler, generator)
        trajectory = self.pipe._last_trajectory  # in HBM; the files are being written behind it

        # Step 2. DDIM sampling + PnP feature and attention injection
        edited_1st_frame = load_image(edited_first_frame_path).resize(tuple(cfg.inverse_config.image_size),
                                                                      resample=Image.Resampling.LANCZOS)
        self.ddim_scheduler.set_timesteps(num_inference_steps)
        t_start = int(self.ddim_scheduler.timesteps[ddim_init_latents_t_idx])
        if t_start not in trajectory:
            raise ValueError(f"no inverted latent at t={t_start}: ddim_inversion_steps={ddim_inversion_steps} and "
                             f"num_inference_steps={num_inference_steps} must produce the same timestep grid")
        ddim_latents_at_t = trajectory[t_start].to(self.device)
        random_latents = torch.randn_like(ddim_latents_at_t)
        rr = cfg.pnp_config.random_ratio
        mixed_latents = random_latents * rr + ddim_latents_at_t * (1 - rr)

        cfg.pnp_config.n_steps = num_inference_steps
        cfg.pnp_config.pnp_f_t = conv_inj
        cfg.pnp_config.pnp_spatial_attn_t = spatial_inj
        cfg.pnp_config.pnp_temp_attn_t = temp_inj
        cfg.pnp_config.ddim_init_latents_t_idx = ddim_init_latents_t_idx
        init_pnp(self.pipe, self.ddim_scheduler, cfg.pnp_config)
        self.pipe.register_modules(scheduler=self.ddim_scheduler)
        edited_video = self.pipe.sample_with_pnp(
            prompt=video_prompt, image=edited_1st_frame, height=cfg.inverse_config.image_size[1],
            width=cfg.inverse_config.image_size[0], num_frames=cfg.inverse_config.n_frames,
            num_inference_steps=cfg.pnp_config.n_steps, guidance_scale=guidance_scale, negative_prompt=video_negative_prompt,
            target_fps=cfg.pnp_config.target_fps, latents=mixed_latents, generator=generator, return_dict=True,
            ddim_init_latents_t_idx=ddim_init_latents_t_idx, ddim_inv_latents_path=trajectory,
            ddim_inv_prompt=cfg.inverse_config.ddim_inv_prompt, ddim_inv_1st_frame=first_frame).frames[0]
        edited_video = [f.resize(tuple(cfg.inverse_config.image_size), resample=Image.LANCZOS) for f in edited_video]
        output_path = os.path.join(tmp_dir, "edited_video.mp4")
        export_to_video(edited_video, output_path, fps=cfg.pnp_config.target_fps)
        trajectory.wait()  # the ddim_latents_{t}.pt files are complete when the call returns
        return output_path
