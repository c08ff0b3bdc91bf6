"""ConsistI2V stage 1 CLI -- DDIM inversion + reconstruction of one clip (flags, config keys and output files of the reference's
``consisti2v/run_ddim_inversion.py``):

    python -m anyv2v_amd.consisti2v_run_ddim_inversion --config configs/consisti2v/pipeline_256/ddim_inversion_256.yaml \
           video_name=clip video_frames_path=/data/clip

Trailing ``key=value`` arguments override the config (``run_ddim_inversion.py:143-150``).
"""
from __future__ import annotations

import argparse
import logging
import os
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from .config import OmegaConf
from .consisti2v_pipeline import ConditionalVideoEditingPipeline, inverse_scheduler_from_pretrained
from .schedulers import DDIMScheduler
from .utils import convert_video_to_frames, export_to_gif, export_to_video, load_ddim_latents_at_t, load_image, seed_everything

MODEL_ID = "TIGER-Lab/ConsistI2V"
logger = logging.getLogger(__name__)


def save_videos_grid(videos: torch.Tensor, path: str, rescale=False, n_rows=6, fps=8, format="gif"):
    """``consisti2v/consisti2v/utils/util.py:21-41``: [b, c, t, h, w] in [0, 1] -> one frame per time step, the b clips side by
    side (``torchvision.utils.make_grid``: 2-pixel black padding, ``n_rows`` per row; a single clip is left as it is), 8-bit by
    TRUNCATION, written as gif or mp4."""
    videos = videos.detach().float().cpu()
    b, c, t, h, w = videos.shape
    frames = []
    for i in range(t):
        x = videos[:, :, i]
        if b == 1:
            grid = x[0]
        else:
            ncol = min(n_rows, b)
            nrow = (b + ncol - 1) // ncol
            grid = torch.zeros(c, nrow * (h + 2) + 2, ncol * (w + 2) + 2)
            for k in range(b):
                r, q = divmod(k, ncol)
                grid[:, r * (h + 2) + 2: r * (h + 2) + 2 + h, q * (w + 2) + 2: q * (w + 2) + 2 + w] = x[k]
        grid = grid.permute(1, 2, 0)
        if rescale:
            grid = (grid + 1.0) / 2.0
        frames.append(Image.fromarray((grid * 255).numpy().astype(np.uint8)))
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    if format == "gif":
        export_to_gif(frames, path, fps=fps)
    elif format == "mp4":
        export_to_video(frames, path, fps=fps)
    else:
        raise ValueError(format)
    return frames


def load_video_frames(frames_path, n_frames):
    """``consisti2v/utils.py:80-84`` (no size check in this backend: the pipeline resizes / crops)."""
    paths = [f"{frames_path}/%05d.png" % i for i in range(n_frames)]
    return paths, [load_image(p) for p in paths]


def frames_of_clip(config):
    """``run_ddim_inversion.py:106-116``: (frame list, frame directory, path of the first frame).  From an mp4 the resized frames go to
    ``<output_dir>/<video name>/`` (``save_dir=config.output_dir``); the first frame is read back from there, so ``save_frames: False``
    with a video file fails here as it does in the reference (no ``00000.png``) -- said up front."""
    if config.get("video_path") and not str(config.video_path).startswith("<") and os.path.isfile(str(config.video_path)):
        save = bool(config.get("save_frames", True))
        if not save:
            raise ValueError("save_frames: False with a video_path: the first frame is opened from <output_dir>/<video name>/00000.png "
                             "(run_ddim_inversion.py:111) -- keep save_frames on, or give video_frames_path")
        frame_list = convert_video_to_frames(config.video_path, tuple(config.image_size), save_frames=save,
                                             save_dir=config.output_dir)[: config.n_frames]
        frames_dir = os.path.join(str(config.output_dir), Path(config.video_path).stem)
        return frame_list, frames_dir, os.path.join(frames_dir, "00000.png")
    if config.get("video_frames_path"):
        _, frame_list = load_video_frames(config.video_frames_path, config.n_frames)
        return frame_list, str(config.video_frames_path), os.path.join(config.video_frames_path, "00000.png")
    raise ValueError("Please provide either video_path or video_frames_path")


def ddim_inversion(config, first_frame, frame_list, pipe: ConditionalVideoEditingPipeline, inverse_scheduler, g):
    """``run_ddim_inversion.py:29-55``."""
    pipe.scheduler = inverse_scheduler
    video_latents_at_0 = pipe.encode_vae_video(frame_list, device=pipe._execution_device, height=config.image_size[1],
                                               width=config.image_size[0])
    ddim_latents = pipe.invert(prompt=config.prompt, first_frame_paths=first_frame, height=config.image_size[1], width=config.image_size[0],
                               video_length=config.n_frames, num_inference_steps=config.n_steps, guidance_scale_txt=config.cfg_txt,
                               guidance_scale_img=config.cfg_img, negative_prompt=config.negative_prompt, frame_stride=config.frame_stride,
                               latents=video_latents_at_0, generator=g, return_dict=False, output_type="latent",
                               output_dir=config.output_dir).videos
    logger.debug(f"ddim_latents.shape: {ddim_latents.shape}")
    return ddim_latents[0]  # [num_inference_steps, c, num_frames, h, w]


def ddim_sampling(config, first_frame, ddim_latents_at_T, pipe: ConditionalVideoEditingPipeline, ddim_scheduler, g, ddim_init_latents_t_idx):
    """``run_ddim_inversion.py:58-77``."""
    pipe.scheduler = ddim_scheduler
    return pipe(prompt=config.prompt, first_frame_paths=first_frame, height=config.image_size[1], width=config.image_size[0],
                video_length=config.n_frames, num_inference_steps=config.n_steps, guidance_scale_txt=config.cfg_txt,
                guidance_scale_img=config.cfg_img, negative_prompt=config.negative_prompt, frame_stride=config.frame_stride,
                latents=ddim_latents_at_T, generator=g, return_dict=True, ddim_init_latents_t_idx=ddim_init_latents_t_idx).videos


def main(config, pipe=None, random_init_seed=None):
    seed_everything(config.seed)
    torch.set_grad_enabled(False)
    device = torch.device(config.device)
    if pipe is None:
        pipe = ConditionalVideoEditingPipeline.from_pretrained(config.get("model_path", MODEL_ID), torch_dtype=torch.float16,
                                                               random_init_seed=random_init_seed)
        pipe.to(device)
    g = torch.Generator().manual_seed(config.seed)
    root = config.get("model_path", MODEL_ID)
    inverse_scheduler = inverse_scheduler_from_pretrained(root)
    ddim_scheduler = DDIMScheduler(**vars(inverse_scheduler.config))
    frame_list, _, first_frame_path = frames_of_clip(config)
    ddim_inversion(config.inverse_config, first_frame_path, frame_list, pipe, inverse_scheduler, g)

    recon_config = config.recon_config
    t_idx = recon_config.ddim_init_latents_t_idx
    ddim_scheduler.set_timesteps(recon_config.n_steps)
    logger.info(f"ddim_scheduler.timesteps: {ddim_scheduler.timesteps}")
    ddim_latents_at_t = load_ddim_latents_at_t(ddim_scheduler.timesteps[t_idx], ddim_latents_path=config.inverse_config.output_dir)
    reconstructed_video = ddim_sampling(recon_config, first_frame_path, ddim_latents_at_t, pipe, ddim_scheduler, g, t_idx)
    os.makedirs(config.output_dir, exist_ok=True)
    save_videos_grid(reconstructed_video, os.path.join(config.output_dir, "ddim_reconstruction.gif"), fps=10, format="gif")
    save_videos_grid(reconstructed_video, os.path.join(config.output_dir, "ddim_reconstruction.mp4"), fps=10, format="mp4")
    logger.info(f"Saved reconstructed video to {config.output_dir}")
    return pipe


def load_config(argv, default):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default=default)
    parser.add_argument("--random_init_seed", type=int, default=None, help="random UNet weights (no checkpoint offline)")
    parser.add_argument("optional_args", nargs="*", default=[])
    args = parser.parse_args(argv)
    config = OmegaConf.load(args.config)
    if args.optional_args:
        config = OmegaConf.merge(config, OmegaConf.from_dotlist(args.optional_args))
    logging.basicConfig(level=logging.DEBUG if config.debug else logging.INFO,
                        format="%(asctime)s - %(levelname)s - [%(funcName)s] - %(message)s")
    logger.info(f"config: {OmegaConf.to_yaml(config)}")
    return config, args


def cli(argv=None):
    config, args = load_config(argv, "configs/consisti2v/pipeline_256/ddim_inversion_256.yaml")
    main(config, random_init_seed=args.random_init_seed)


if __name__ == "__main__":
    cli()
