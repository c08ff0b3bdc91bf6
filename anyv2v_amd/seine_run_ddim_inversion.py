"""SEINE stage 1 CLI -- DDIM inversion + reconstruction of one clip (flags, config keys and output files of the reference's
``seine/run_ddim_inversion.py``):

    python -m anyv2v_amd.seine_run_ddim_inversion --config configs/seine/ddim_inversion.yaml --video_path /data/clip.mp4 \
           [--gpu 0] [--width 512 --height 320]

Writes ``<output_dir>/seine/<clip>/steps_<n>/nframes_<f>/`` with ``ddim_latents/ddim_latents_{t}.pt`` (the ``n_save_steps`` DDIM
timesteps), ``inversion_prompts.yaml``, ``config.yaml``, ``recon_frames/%05d.png`` and ``inverted.mp4``.  ``--video_path`` may also be a
directory of ``%05d.png`` frames (no video decoder library is available here for arbitrary mp4 files).
"""
from __future__ import annotations

import argparse
import logging
import os
from pathlib import Path

import torch
import yaml
from PIL import Image

from .config import OmegaConf
from .schedulers import DDIMScheduler
from .seine_pipeline import SEINEDDIMInversionPipeline, _scheduler
from .utils import convert_video_to_frames, export_to_video, seed_everything

logger = logging.getLogger(__name__)


def add_dict_to_yaml_file(file_path, key, value):
    """``run_ddim_inversion.py:35-46``."""
    data = {}
    if os.path.exists(file_path):
        with open(file_path, "r") as f:
            data = yaml.safe_load(f) or {}
    data[key] = value
    with open(file_path, "w") as f:
        yaml.dump(data, f)


def get_timesteps(scheduler, num_inference_steps, strength):
    """``run_ddim_inversion.py:49-56``."""
    init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
    t_start = max(num_inference_steps - init_timestep, 0)
    return scheduler.timesteps[t_start:], num_inference_steps - t_start


def save_frames_png_and_mp4(frames_u8, frames_dir, video_path, fps=8):
    """uint8 [f, c, h, w] -> ``frames_dir/%05d.png`` + an mp4 (``run_ddim_inversion.py:318-326``)."""
    os.makedirs(frames_dir, exist_ok=True)
    pil = [Image.fromarray(fr.permute(1, 2, 0).cpu().numpy()) for fr in frames_u8]
    for i, im in enumerate(pil):
        im.save(os.path.join(frames_dir, f"{i:05d}.png"))
    export_to_video(pil, video_path, fps=fps)


def main(config, device, pipeline=None, random_init_seed=None):
    assert config.model_name == "seine", f"model_name {config.model_name} not supported."
    toy_scheduler = _scheduler(config, DDIMScheduler)
    toy_scheduler.set_timesteps(config.n_save_steps)
    timesteps_to_save, num_inference_steps = get_timesteps(toy_scheduler, num_inference_steps=config.n_save_steps, strength=1.0)
    logger.info(f"timesteps_to_save: {timesteps_to_save}")
    save_path = os.path.join(config.output_dir, config.model_name, Path(config.src_video_path).stem, f"steps_{config.n_steps}",
                             f"nframes_{config.n_frame_to_invert}")
    logger.info(f"save_path: {save_path}")
    os.makedirs(os.path.join(save_path, "ddim_latents"), exist_ok=True)
    add_dict_to_yaml_file(file_path=os.path.join(save_path, "inversion_prompts.yaml"), key=Path(config.src_video_path).stem,
                          value=config.inversion_prompt)
    with open(os.path.join(save_path, "config.yaml"), "w") as f:
        yaml.dump(OmegaConf.to_container(config, resolve=True), f)
    pipe = pipeline if pipeline is not None else SEINEDDIMInversionPipeline(device, config, random_init_seed=random_init_seed)
    recon_frames = pipe.extract_ddim_latents(config, timesteps_to_save, save_path)          # [1, f, h, w, c]
    recon_frames = recon_frames[0].permute(0, 3, 1, 2)
    save_frames_png_and_mp4(recon_frames, os.path.join(save_path, "recon_frames"), os.path.join(save_path, "inverted.mp4"), fps=8)
    logger.info(f"Saved reconstructed frames and video to {save_path}")
    return save_path


def cli(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default="./configs/seine/ddim_inversion.yaml")
    parser.add_argument("--video_path", type=str, required=False, help="Path to the video to invert.")
    parser.add_argument("--gpu", type=int, required=False, help="GPU number to use.")
    parser.add_argument("--width", type=int, required=False)
    parser.add_argument("--height", type=int, required=False)
    parser.add_argument("--random_init_seed", type=int, default=None, help="random UNet weights (no checkpoint offline)")
    parser.add_argument("optional_args", nargs="*", default=[])
    args = parser.parse_args(argv)
    config = OmegaConf.load(args.config)
    if args.optional_args:
        config = OmegaConf.merge(config, OmegaConf.from_dotlist(args.optional_args))
    if args.video_path is not None:
        config.src_video_path = args.video_path
    if args.gpu is not None:
        config.device = f"cuda:{args.gpu}"
    if args.width is not None and args.height is not None:
        config.image_size = [args.height, args.width]
    logging.basicConfig(level=logging.DEBUG if config.debug else logging.INFO,
                        format="%(asctime)s - %(levelname)s - [%(funcName)s] - %(message)s")
    logger.info(f"config: {config}")
    assert os.path.exists(config.src_video_path), f"src_video_path {config.src_video_path} does not exist."
    if os.path.isfile(str(config.src_video_path)):   # save_video_as_frames (``pnp_utils.py:30-43``): <dir>/<stem>/%05d.png, LANCZOS to (w, h)
        convert_video_to_frames(str(config.src_video_path), (config.image_size[1], config.image_size[0]), save_frames=True)
        config.src_video_path = os.path.join(Path(config.src_video_path).parent, Path(config.src_video_path).stem)
    device = torch.device(config.device)
    torch.set_grad_enabled(False)
    seed_everything(config.seed)
    return main(config, device, random_init_seed=args.random_init_seed)


if __name__ == "__main__":
    cli()
