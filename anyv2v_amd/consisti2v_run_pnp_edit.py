"""ConsistI2V stage 2 CLI -- PnP edit of one inverted clip (flags, config keys and output files of the reference's
``consisti2v/run_pnp_edit.py``):

    python -m anyv2v_amd.consisti2v_run_pnp_edit --config configs/consisti2v/pipeline_256/pnp_edit.yaml video_name=clip \
           video_frames_path=/data/clip edited_first_frame_path=/data/clip_edit.png ddim_latents_path=outputs editing_prompt="..."
"""
from __future__ import annotations

import logging
import os
from pathlib import Path

import torch

from . import consisti2v as c2
from .consisti2v_pipeline import ConditionalVideoEditingPipeline, inverse_scheduler_from_pretrained
from .consisti2v_run_ddim_inversion import MODEL_ID, load_config, load_video_frames, save_videos_grid
from .schedulers import DDIMScheduler
from .utils import convert_video_to_frames, load_ddim_latents_at_t, seed_everything

logger = logging.getLogger(__name__)


def init_pnp(pipe, scheduler, config):
    """``run_pnp_edit.py:31-47``: the first ``int(n_steps * ratio)`` sampling timesteps of each site family."""
    conv_injection_t = int(config.n_steps * config.pnp_f_t)
    spatial_attn_qk_injection_t = int(config.n_steps * config.pnp_spatial_attn_t)
    temp_attn_qk_injection_t = int(config.n_steps * config.pnp_temp_attn_t)
    pick = lambda n: scheduler.timesteps[:n] if n >= 0 else []
    c2.register_conv_injection(pipe, pick(conv_injection_t))
    c2.register_spatial_attention_pnp(pipe, pick(spatial_attn_qk_injection_t))
    c2.register_temp_attention_pnp(pipe, pick(temp_attn_qk_injection_t))
    logger.debug(f"conv_injection_t: {conv_injection_t}")
    logger.debug(f"spatial_attn_qk_injection_t: {spatial_attn_qk_injection_t}")
    logger.debug(f"temp_attn_qk_injection_t: {temp_attn_qk_injection_t}")


def main(config, pipe=None, random_init_seed=None):
    device = torch.device(config.device)
    seed_everything(config.seed)
    torch.set_grad_enabled(False)
    root = config.get("model_path", MODEL_ID)
    if pipe is None:
        pipe = ConditionalVideoEditingPipeline.from_pretrained(root, torch_dtype=torch.float16, random_init_seed=random_init_seed)
        pipe.to(device)
    ddim_scheduler = DDIMScheduler(**vars(inverse_scheduler_from_pretrained(root).config))
    if config.get("video_path") and not str(config.video_path).startswith("<") and os.path.isfile(str(config.video_path)):
        convert_video_to_frames(config.video_path, tuple(config.image_size), save_frames=True)
        config.video_frames_path = f"{Path(config.video_path).parent}/{Path(config.video_path).stem}"
    elif config.get("video_frames_path"):
        load_video_frames(config.video_frames_path, config.n_frames)
    else:
        raise ValueError("Please provide either video_path or video_frames_path")
    src_1st_frame = os.path.join(config.video_frames_path, "00000.png")
    edited_1st_frame = config.edited_first_frame_path

    t_idx = config.ddim_init_latents_t_idx
    ddim_scheduler.set_timesteps(config.n_steps)
    logger.info(f"ddim_scheduler.timesteps: {ddim_scheduler.timesteps}")
    ddim_latents_path = os.path.join(config.ddim_latents_path, config.exp_name)
    ddim_latents_at_t = load_ddim_latents_at_t(ddim_scheduler.timesteps[t_idx], ddim_latents_path=ddim_latents_path)
    logger.debug(f"ddim_latents_at_t.shape: {ddim_latents_at_t.shape}")
    # blend with fresh noise (``run_pnp_edit.py:93-96``; global RNG, seeded above)
    random_latents = torch.randn_like(ddim_latents_at_t.float())
    mixed_latents = random_latents * config.blend_ratio + ddim_latents_at_t.float() * (1 - config.blend_ratio)

    init_pnp(pipe, ddim_scheduler, config)
    pipe.register_modules(scheduler=ddim_scheduler)
    edited_video = pipe.sample_with_pnp(
        prompt=config.editing_prompt, first_frame_paths=edited_1st_frame, height=config.image_size[1], width=config.image_size[0],
        video_length=config.n_frames, num_inference_steps=config.n_steps, guidance_scale_txt=config.cfg_txt, guidance_scale_img=config.cfg_img,
        negative_prompt=config.editing_negative_prompt, frame_stride=config.frame_stride, latents=mixed_latents,
        generator=torch.manual_seed(config.seed), return_dict=True, ddim_init_latents_t_idx=t_idx, ddim_inv_latents_path=ddim_latents_path,
        ddim_inv_prompt=config.ddim_inv_prompt, ddim_inv_1st_frame_path=src_1st_frame).videos
    os.makedirs(config.output_dir, exist_ok=True)
    save_videos_grid(edited_video, os.path.join(config.output_dir, config.editing_prompt, "video.gif"), fps=8, format="gif")
    save_videos_grid(edited_video, os.path.join(config.output_dir, config.editing_prompt, "video.mp4"), fps=8, format="mp4")
    logger.info(f"Saved edited video to {config.output_dir}")
    return edited_video


def cli(argv=None):
    config, args = load_config(argv, "configs/consisti2v/pipeline_256/pnp_edit.yaml")
    main(config, random_init_seed=args.random_init_seed)


if __name__ == "__main__":
    cli()
