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


def main_pipelined(inv_template, inv_list, edit_template, edit_list, device, logger, pipe, shares=None):
    """One GPU, several clips: clip k + 1 is INVERTED while clip k is EDITED.  The two loops have no data in common (the edit reads
    the trajectory of its own clip, complete before it starts), so they run on two HIP streams -- stage 1 on ``pipe``, stage 2 on
    ``pipe.sibling()`` (same weights; own scheduler slot, step engines, graphs and split-K scratch) -- and the launches that do not
    fill the chip on their own (the low-resolution levels, the B = 1 inversion) run side by side: 105.8 -> 98.8 ms per step pair at
    16 f x 512^2 (``bench.py``), latents bit-equal to the serial order.

    Host order per clip k: [first edit of clip k: 50 steps enqueued | stream B, behind the event recorded after inversion k]
    -> [inversion k + 1: frames, VAE encode, steps enqueued | stream A] -> [edit k: decode, files; further edits of clip k | B]
    -> [inversion k + 1: files, reconstruction | A].  When the inversion has no more steps than the edit (the 50 + 50 case), its
    step i is held back until edit step i * n_edit / n_inv starts (stream events), so that the two loops stay side by side over
    the whole edit instead of the inversion racing ahead; a longer inversion (the template's 500 steps) is never held back (the
    edit is then a fifth of the clip's work and the job gains ~2 %: only the overlapped part runs two-streamed).  Every entry re-seeds the RNGs when it starts and a parked entry gets its RNG state
    back before it finishes, so every file equals the serial order's."""
    import contextlib

    from . import ops
    cuda = device.type == "cuda"
    stream_a = torch.cuda.Stream(device) if cuda else None
    stream_b = torch.cuda.Stream(device) if cuda else None
    on = (lambda st: torch.cuda.stream(st)) if cuda else (lambda st: contextlib.nullcontext())
    trajectories = {}
    inv_mine, edit_mine = shares if shares is not None else (None, None)
    s1 = stage1.Stage1(inv_template, inv_list, device, logger, pipe=pipe, trajectories=trajectories, my_entries=inv_mine)
    pipe_b = pipe.sibling(ws_slot=1)
    s2 = stage2.Stage2(edit_template, edit_list, device, logger, pipe=pipe_b, trajectories=trajectories, my_entries=edit_mine)
    if cuda:
        stream_a.wait_stream(torch.cuda.current_stream(device))
        stream_b.wait_stream(torch.cuda.current_stream(device))
    pending = list(s2.my_entries)
    pace = []   # one event per edit step of the edit that is in flight (stream B)

    def record(i, n):
        if i == 0:
            pace.clear()
        pace.append(stream_b.record_event())

    def wait(i, n):
        if pace and n <= len(pace):
            stream_a.wait_event(pace[i * len(pace) // n])
    if cuda and os.environ.get("ANYV2V_PIPELINE_PACING", "1") == "1":
        pipe_b.pace_record, pipe.pace_wait = record, wait

    def take(latents_dir):
        mine = [e for e in pending if s2.latents_dir_of(e) == latents_dir]
        for e in mine:
            pending.remove(e)
        return mine

    def start_edit(e, ready):
        with on(stream_b), ops.workspace_slot(1):
            if ready is not None:
                stream_b.wait_event(ready)
            g = s2.entry(e)
            next(g, None)
        return g

    def finish_edits(g, rest):
        with on(stream_b), ops.workspace_slot(1):
            for _ in g:
                pass
            pace.clear()                             # (further edits of the clip run un-paced beside nothing)
            for e in rest:
                s2.run_entry(e)

    prev = None   # (latents directory, event) of the last inversion that was run here
    for entry in s1.entries():
        mine = take(prev[0]) if prev is not None else []
        g = start_edit(mine[0], prev[1]) if mine else None          # edit of clip k: enqueued, pace events recorded
        with on(stream_a):
            latents_dir = next(entry, None)                          # inversion k + 1: enqueued beside it
            ready = stream_a.record_event() if (cuda and latents_dir is not None) else None
        if g is not None:
            finish_edits(g, mine[1:])
        with on(stream_a):
            for _ in entry:                                          # files, reconstruction
                pass
        if latents_dir is not None:
            prev = (latents_dir, ready)
    pace.clear()
    with on(stream_b), ops.workspace_slot(1):
        if prev is not None and cuda:
            stream_b.wait_event(prev[1])
        for e in (take(prev[0]) if prev is not None else []) + list(pending):   # last clip; then entries inverted elsewhere (files)
            if e in pending:
                pending.remove(e)
            s2.run_entry(e)
    pipe_b.pace_record = pipe.pace_wait = None
    if cuda:
        torch.cuda.current_stream(device).wait_stream(stream_a)
        torch.cuda.current_stream(device).wait_stream(stream_b)
    s2.finish()
    return trajectories


def main(inv_template, inv_list, edit_template, edit_list, device, logger, synthetic_encoders=False, random_init_seed=None,
         frame_parallel=False, pipelined=None, batch_clips=1):
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
    if batch_clips and int(batch_clips) > 1:
        pipelined = False   # batched inversions (Stage1.run_batched) run stage 1 over all clips first; the edits follow from HBM
    if pipelined and world == 1:
        return main_pipelined(inv_template, inv_list, edit_template, edit_list, device, logger, pipe)
    n_clips = len({e.get("video_name") for e in inv_list if e.get("active", True) is not False})
    if pipelined and world > 1 and not frame_parallel and n_clips >= world:
        # enough clips for every rank: deal whole clips (an inversion and its edits) instead of the entries of each stage, and let
        # every rank pipeline its own clips; no rank waits for another's files, so the barrier between the stages is not needed
        from .parallel import shard_by_clip
        return main_pipelined(inv_template, inv_list, edit_template, edit_list, device, logger, pipe,
                              shares=shard_by_clip(inv_list, edit_list, rank, world))
    trajectories = {}
    seed_everything(inv_template.seed)  # each stage starts from its template's seed, as two separate processes would
    stage1.main(inv_template, inv_list, device, logger, synthetic_encoders, random_init_seed, frame_parallel, pipe=pipe,
                trajectories=trajectories, batch_clips=batch_clips)
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
    ap.add_argument("--batch_clips", type=int, default=1,
                    help="invert up to N clips of the same geometry in one batch (for inversion-bound jobs, e.g. the template's 500 steps; "
                         "implies stage 1 before stage 2)")
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
         pipelined=False if args.serial else None, batch_clips=args.batch_clips)


if __name__ == "__main__":
    cli()
