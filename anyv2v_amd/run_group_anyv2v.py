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


def main_pipelined(inv_template, inv_list, edit_template, edit_list, device, logger, pipe):
    """One GPU, several clips: clip k + 1 is INVERTED while clip k is EDITED.  The two loops have no data in common (the edit reads
    the trajectory of its own clip, complete before it starts), so they run on two HIP streams -- stage 1 on ``pipe``, stage 2 on
    ``pipe.sibling()`` (same weights; own scheduler slot, step engines, graphs and split-K scratch) -- and the launches that do not
    fill the chip on their own (the low-resolution levels, the B = 1 inversion) run side by side: 106.4 -> 99.7 ms per step pair at
    16 f x 512^2 (``bench.py --overlap``), latents bit-equal to the serial order.

    Host order per clip k: [enqueue inversion k + 1 | stream A] -> [edits of clip k, files | stream B, after the event recorded
    behind inversion k] -> [files + reconstruction of clip k + 1 | stream A].  Every entry re-seeds the RNGs when it starts and the
    parked inversion entry gets its RNG state back before it finishes, so every file equals the two-stage run's."""
    import contextlib

    from . import ops
    cuda = device.type == "cuda"
    stream_a = torch.cuda.Stream(device) if cuda else None
    stream_b = torch.cuda.Stream(device) if cuda else None
    on = (lambda st: torch.cuda.stream(st)) if cuda else (lambda st: contextlib.nullcontext())
    trajectories = {}
    s1 = stage1.Stage1(inv_template, inv_list, device, logger, pipe=pipe, trajectories=trajectories)
    s2 = stage2.Stage2(edit_template, edit_list, device, logger, pipe=pipe.sibling(ws_slot=1), trajectories=trajectories)
    if cuda:
        stream_a.wait_stream(torch.cuda.current_stream(device))
        stream_b.wait_stream(torch.cuda.current_stream(device))
    pending = list(s2.my_entries)

    def edits_of(latents_dir, ready):
        """The edit entries of this clip, list order; stream B first waits for the clip's inversion."""
        mine = [e for e in pending if s2.latents_dir_of(e) == latents_dir]
        if not mine:
            return
        with on(stream_b), ops.workspace_slot(1):
            if ready is not None:
                stream_b.wait_event(ready)
            for e in mine:
                pending.remove(e)
                s2.run_entry(e)

    prev = None   # (latents directory, event) of the inversion that is complete or running ahead of the edits
    for entry in s1.entries():
        with on(stream_a):
            latents_dir = next(entry, None)          # frames, VAE encode, inversion steps: launched, nothing read back
            ready = stream_a.record_event() if (cuda and latents_dir is not None) else None
        if prev is not None:
            edits_of(*prev)                          # ... while the previous clip is edited
        with on(stream_a):
            for _ in entry:                          # files, reconstruction
                pass
        prev = (latents_dir, ready) if latents_dir is not None else None
    if prev is not None:
        edits_of(*prev)
    with on(stream_b), ops.workspace_slot(1):        # entries whose inversion was not run here (complete on disk): from the files
        for e in list(pending):
            pending.remove(e)
            s2.run_entry(e)
    if cuda:
        torch.cuda.current_stream(device).wait_stream(stream_a)
        torch.cuda.current_stream(device).wait_stream(stream_b)
    s2.finish()
    return trajectories


def main(inv_template, inv_list, edit_template, edit_list, device, logger, synthetic_encoders=False, random_init_seed=None,
         frame_parallel=False, pipelined=None):
    rank, local_rank, world = init_distributed()
    pipe = I2VGenXLPipeline.from_pretrained(inv_template.get("model_path", stage1.MODEL_ID), torch_dtype=torch.float16,
                                            variant="fp16", random_init_seed=random_init_seed)
    pipe.to(device)
    if synthetic_encoders:
        attach_synthetic_encoders(pipe)
    if frame_parallel and world > 1:
        pipe.unet.set_frame_parallel(FrameParallel())
    if pipelined is None:
        pipelined = os.environ.get("ANYV2V_PIPELINED", "1") == "1"
    if pipelined and world == 1:
        return main_pipelined(inv_template, inv_list, edit_template, edit_list, device, logger, pipe)
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
    ap.add_argument("--serial", action="store_true", help="stage 1 over all clips, then stage 2 (default on one GPU: clip k + 1 is "
                                                          "inverted while clip k is edited, on two streams)")
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
    main(inv_t, inv_l, ed_t, ed_l, device, logger, args.synthetic_encoders, args.random_init_seed, args.frame_parallel,
         pipelined=False if args.serial else None)


if __name__ == "__main__":
    cli()
