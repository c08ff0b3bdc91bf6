"""Does the PHASE between the two loops of the clip pipeline matter?  `bench.py`'s pipelined step pair (an edit step on one stream, an inversion
step of the next clip on another, the inversion step released by an event recorded when the edit step starts) with the inversion step held
back by a further X ms (a spin kernel on its stream): both forwards are high-resolution at their ends and low-resolution in the middle, so
X decides which parts meet.  `python tools/phase_shift_probe.py [X ms ...]` -> one JSON line per X (interleaved repeats)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from anyv2v_amd import pnp_utils  # noqa: E402
from anyv2v_amd.pipeline import I2VGenXLPipeline, _StepEngine  # noqa: E402
from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler  # noqa: E402


def main():
    shifts = [float(a) for a in sys.argv[1:]] or [0, 10, 20, 30, 40, 50]
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    pipe = I2VGenXLPipeline.from_pretrained("ali-vilab/i2vgen-xl", torch_dtype=torch.float16, variant="fp16", random_init_seed=0).to(dev)
    n = bench.STEPS_PER_STAGE
    inv, fwd = DDIMInverseScheduler(), DDIMScheduler()
    inv.set_timesteps(n)
    fwd.set_timesteps(n)
    ts_inv, ts_pnp = [int(t) for t in inv.timesteps], [int(t) for t in fwd.timesteps]
    for reg in (pnp_utils.register_conv_injection, pnp_utils.register_spatial_attention_pnp, pnp_utils.register_temp_attention_pnp):
        reg(pipe, fwd.timesteps)
    tt_inv = torch.tensor(ts_inv, dtype=torch.float32, device=dev)[:, None].contiguous()
    tt_pnp = torch.tensor(ts_pnp, dtype=torch.float32, device=dev)[:, None].expand(-1, 3).contiguous()
    cf_inv, cf_pnp = inv.coefficient_table(ts_inv, dev), fwd.coefficient_table(ts_pnp, dev)
    fps1, fps3 = torch.tensor([8], device=dev), torch.tensor([8, 8, 8], device=dev)
    lat, ehs, ie, il_all = bench.synthetic_clip(dev, 8888)
    s_inv, s_pnp = lat.clone(), lat.repeat(3, 1, 1, 1, 1).contiguous()
    cond1 = dict(encoder_hidden_states=ehs[:1].contiguous(), fps=fps1, image_latents=il_all[:1].contiguous(), image_embeddings=ie[:1].contiguous())
    cond3 = dict(encoder_hidden_states=ehs, fps=fps3, image_latents=il_all, image_embeddings=ie)
    pnp_utils.clear_time(pipe)
    e_inv = _StepEngine(pipe, s_inv, cond1, b_unc=-1, b_cond=0, guidance=1.0, dup_slots=[])
    e_pnp = _StepEngine(pipe.sibling(ws_slot=1), s_pnp, cond3, b_unc=1, b_cond=2, guidance=9.0, dup_slots=[1], shared_stem=True)
    e_pnp.drop_src_tail = True
    traj = torch.zeros(n, 4, bench.FRAMES, bench.LAT, bench.LAT, dtype=torch.float16, device=dev)
    for i in range(n):
        e_inv.step(tt_inv[i], cf_inv[i], key=("inv",))
        traj[i].copy_(s_inv[0])
    s_inv.copy_(lat)
    st_inv, st_pnp = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    # spin-kernel calibration: cycles per millisecond
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000000)
    torch.cuda.synchronize()
    e0.record()
    torch.cuda._sleep(20000000)
    e1.record()
    torch.cuda.synchronize()
    cyc_per_ms = 20000000 / e0.elapsed_time(e1)
    print(json.dumps(dict(spin_cycles_per_ms=round(cyc_per_ms))), flush=True)

    def pair(i, shift_ms):
        j = i % n
        with torch.cuda.stream(st_pnp):
            started = st_pnp.record_event()
            s_pnp[0].copy_(traj[j])
            pnp_utils.register_time(pipe, ts_pnp[j])
            e_pnp.step(tt_pnp[j], cf_pnp[j], key=("pnp",) + pnp_utils.injection_state(pipe))
        with torch.cuda.stream(st_inv):
            st_inv.wait_event(started)
            if shift_ms > 0:
                torch.cuda._sleep(int(shift_ms * cyc_per_ms))
            pnp_utils.clear_time(pipe)
            e_inv.step(tt_inv[j], cf_inv[j], key=("inv",))
    res = {x: [] for x in shifts}
    for rep in range(2):
        for x in shifts:
            for i in range(3):
                pair(i, x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20):
                pair(i, x)
            torch.cuda.synchronize()
            res[x].append((time.perf_counter() - t0) / 20 * 1e3)
    for x in shifts:
        print(json.dumps(dict(inversion_held_back_ms=x, ms_per_step_pair=[round(v, 3) for v in res[x]])), flush=True)


if __name__ == "__main__":
    main()
