"""SEINE stage 2 CLI -- PnP edit of one inverted clip (flags, config keys and output files of the reference's ``seine/run_pnp_edit.py``):

    python -m anyv2v_amd.seine_run_pnp_edit --config configs/seine/pnp_edit.yaml src_video_path=/data/clip.mp4 \
           edited_first_frame_path=/data/clip_edit.png prompt="..." ddim_inversion_dir=ddim-inversion/default

Writes ``<output_dir>/seine/<clip>/<prompt_with_underscores>/cfg.._f.._spa.._cro.._tmp.._stp../img_ode/%05d.png`` and
``video_pnp_fps_8.mp4``.
"""
from __future__ import annotations

import argparse
import logging
import os
from pathlib import Path

import torch

from .config import OmegaConf
from .seine_pipeline import SEINEPnPPipeline
from .seine_run_ddim_inversion import save_frames_png_and_mp4
from .utils import seed_everything

logger = logging.getLogger(__name__)


def main(config, device, pipeline=None, random_init_seed=None):
    """``run_pnp_edit.py:363-393``."""
    save_path = os.path.join(
        config.output_dir, config.model_name, Path(config.src_video_path).stem, config.prompt.replace(" ", "_")[:240],
        f"cfg{config.cfg_scale}_f{config.pnp_f_t}_spa{config.pnp_spatial_attn_t}_cro{config.pnp_cross_attn_t}_tmp{config.pnp_temp_attn_t}_stp{config.n_steps}"
        if config.enable_pnp else "")
    config.output_path = save_path
    logger.info(f"save_path: {save_path}")
    pipe = pipeline if pipeline is not None else SEINEPnPPipeline(device, config, random_init_seed=random_init_seed)
    pipe.scheduler.set_timesteps(config.n_steps)
    if config.enable_pnp:
        pipe.scheduler.set_timesteps(config.n_steps)
        pipe.init_pnp()
    edited_frames = pipe.edit_video(config)[0].permute(0, 3, 1, 2)     # [f, c, h, w] uint8
    save_frames_png_and_mp4(edited_frames, f"{config.output_path}/img_ode", f"{config.output_path}/video_pnp_fps_8.mp4", fps=8)
    logger.info(f"Saved video to {config.output_path}")
    return save_path


def cli(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default="./configs/seine/pnp_edit.yaml")
    parser.add_argument("--random_init_seed", type=int, default=None, help="random UNet weights (no checkpoint offline)")
    parser.add_argument("optional_args", nargs="*", default=[])
    args = parser.parse_args(argv)
    config = OmegaConf.load(args.config)
    if args.optional_args:
        config = OmegaConf.merge(config, OmegaConf.from_dotlist(args.optional_args))
    logging.basicConfig(level=logging.DEBUG if config.debug else logging.INFO,
                        format="%(asctime)s - %(levelname)s - [%(funcName)s] - %(message)s")
    logger.info(f"config: {config}")
    device = torch.device(config.device)
    torch.set_grad_enabled(False)
    seed_everything(config.seed)
    return main(config, device, random_init_seed=args.random_init_seed)


if __name__ == "__main__":
    cli()
