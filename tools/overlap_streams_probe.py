"""How much more does the chip take?  K independent "lanes", each the bench's pipelined step pair (an inversion step of one clip beside an
edit step of another, two streams, own engines and scratch slots), enqueued side by side: ms per step pair = wall / (pairs x K).
K = 1 is `bench.py`'s default schedule; K = 2 asks whether a job with two clips in flight per stage would gain anything.
`python tools/overlap_streams_probe.py [K ...]` -> one JSON line per K."""
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
    ks = [int(a) for a in sys.argv[1:]] or [1, 2]
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
    lanes = []
    for k in range(max(ks)):
        lat, ehs, ie, il_all = bench.synthetic_clip(dev, 8888 + k)
        s_inv, s_pnp = lat.clone(), lat.repeat(3, 1, 1, 1, 1).contiguous()
        cond1 = dict(encoder_hidden_states=ehs[:1].contiguous(), fps=fps1, image_latents=il_all[:1].contiguous(), image_embeddings=ie[:1].contiguous())
        cond3 = dict(encoder_hidden_states=ehs, fps=fps3, image_latents=il_all, image_embeddings=ie)
        pnp_utils.clear_time(pipe)
        e_inv = _StepEngine(pipe.sibling(ws_slot=2 * k), s_inv, cond1, b_unc=-1, b_cond=0, guidance=1.0, dup_slots=[])
        e_pnp = _StepEngine(pipe.sibling(ws_slot=2 * k + 1), s_pnp, cond3, b_unc=1, b_cond=2, guidance=9.0, dup_slots=[1], shared_stem=True)
        e_pnp.drop_src_tail = True
        traj = torch.zeros(n, 4, bench.FRAMES, bench.LAT, bench.LAT, dtype=torch.float16, device=dev)
        for i in range(n):          # the trajectory the edit reads (an earlier inversion of its clip)
            e_inv.step(tt_inv[i], cf_inv[i], key=("inv",))
            traj[i].copy_(s_inv[0])
        s_inv.copy_(lat)
        lanes.append(dict(e_inv=e_inv, e_pnp=e_pnp, s_inv=s_inv, s_pnp=s_pnp, traj=traj, st_inv=torch.cuda.Stream(), st_pnp=torch.cuda.Stream()))
    torch.cuda.synchronize()

    def pair(i, K):
        j = i % n
        for L in lanes[:K]:
            with torch.cuda.stream(L["st_pnp"]):
                L["s_pnp"][0].copy_(L["traj"][j])
                pnp_utils.register_time(pipe, ts_pnp[j])
                L["e_pnp"].step(tt_pnp[j], cf_pnp[j], key=("pnp",) + pnp_utils.injection_state(pipe))
            with torch.cuda.stream(L["st_inv"]):
                pnp_utils.clear_time(pipe)
                L["e_inv"].step(tt_inv[j], cf_inv[j], key=("inv",))
    for K in ks:
        for i in range(2):
            pair(i, K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(30):
            pair(i, K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps(dict(lanes=K, streams=2 * K, ms_per_step_pair=round(dt / (30 * K) * 1e3, 3),
                              finite=bool(all(torch.isfinite(L["s_pnp"].float()).all() for L in lanes[:K])))))


if __name__ == "__main__":
    main()
