"""Both stages of the reference in ONE process: ``run_group_ddim_inversion`` then ``run_group_pnp_edit`` over the same group
files, one pipeline (weights loaded once), the inversion trajectory of each clip handed to its edits in HBM (SURVEY.md 8(f) F2:
"fusing stage 1 -> stage 2 in one process removes 500 file round-trips per clip").  The reference does this only in its
front-ends (``gradio_demo.py:58-222``, ``predict.py:43-258``); its two CLI stages communicate through ``ddim_latents_{t}.pt``.
Every file of both stages is still written (same names, same formats), so the outputs are interchangeable with the two-step
run; an edit entry whose inversion was skipped here (directory already complete) reads the files as stage 2 alone would.

    python -m anyv2v_amd.run_group_anyv2v --inversion_template configs/group_ddim_inversion/template.yaml \\
        --inversion_configs_json configs/group_ddim_inversion/group_config.json \\
        --edit_template configs/group_pnp_edit/template.yaml --edit_configs_json configs/group_pnp_edit/group_config.json
"""
from __future__ import annotations

import argparse
import json
import logging
import os

import torch

from . import run_group_ddim_inversion as stage1
from . import run_group_pnp_edit as stage2
from .config import OmegaConf
from .encoders import attach_synthetic_encoders
from .parallel import FrameParallel, init_distributed
from .pipeline import I2VGenXLPipeline
from .utils import seed_everything


def main(inv_template, inv_list, edit_template, edit_list, device, logger, synthetic_encoders=False, random_init_seed=None,
         frame_parallel=False):
    rank, local_rank, world = init_distributed()
    pipe = I2VGenXLPipeline.from_pretrained(inv_template.get("model_path", stage1.MODEL_ID), torch_dtype=torch.float16,
                                            variant="fp16", random_init_seed=random_init_seed)
    pipe.to(device)
    if synthetic_encoders:
        attach_synthetic_encoders(pipe)
    if frame_parallel and world > 1:
        pipe.unet.set_frame_parallel(FrameParallel())
    trajectories = {}
    seed_everything(inv_template.seed)  # each stage starts from its template's seed, as two separate processes would
    stage1.main(inv_template, inv_list, device, logger, synthetic_encoders, random_init_seed, frame_parallel, pipe=pipe,
                trajectories=trajectories)
    if world > 1 and not frame_parallel:
        # Sharded mode deals the entries of each stage independently: a rank may edit a clip that another rank inverted (or have no
        # inversion work at all) and then reads its ddim_latents files -- which must be complete on disk first.  Join this rank's
        # background writers, then wait for every rank (the two-step CLI gets this for free: stage 1 exits before stage 2 starts).
        import torch.distributed as dist
        from .utils import wait_for_pending_writes
        wait_for_pending_writes()
        dist.barrier()
    seed_everything(edit_template.seed)
    stage2.main(edit_template, edit_list, device, logger, synthetic_encoders, random_init_seed, frame_parallel, pipe=pipe,
                trajectories=trajectories)
    return trajectories


def cli(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--inversion_template", type=str, default="./configs/group_ddim_inversion/template.yaml")
    ap.add_argument("--inversion_configs_json", type=str, default="./configs/group_ddim_inversion/group_config.json")
    ap.add_argument("--edit_template", type=str, default="./configs/group_pnp_edit/template.yaml")
    ap.add_argument("--edit_configs_json", type=str, default="./configs/group_pnp_edit/group_config.json")
    ap.add_argument("--synthetic_encoders", action="store_true")
    ap.add_argument("--random_init_seed", type=int, default=None)
    ap.add_argument("--frame_parallel", action="store_true")
    args = ap.parse_args(argv)
    inv_t, ed_t = OmegaConf.load(args.inversion_template), OmegaConf.load(args.edit_template)
    logging.basicConfig(level=logging.DEBUG if inv_t.debug else logging.INFO,
                        format="%(asctime)s - %(levelname)s - [%(funcName)s] - %(message)s")
    logger = logging.getLogger(__name__)
    inv_l, ed_l = json.load(open(args.inversion_configs_json)), json.load(open(args.edit_configs_json))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = inv_t.device if world == 1 else f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
    device = torch.device(dev)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    torch.set_grad_enabled(False)
    main(inv_t, inv_l, ed_t, ed_l, device, logger, args.synthetic_encoders, args.random_init_seed, args.frame_parallel)


if __name__ == "__main__":
    cli()
