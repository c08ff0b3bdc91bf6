"""N bench step pairs (1 inversion step B=1 + 1 PnP edit step B=3, eager launches) and nothing else, for whole-step
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (tools/step_traffic.sh).  python tools/step_traffic_target.py [pairs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ANYV2V_NO_GRAPH"] = "1"
import torch  # noqa: E402

import bench  # noqa: E402
from anyv2v_amd import pnp_utils  # noqa: E402
from anyv2v_amd.pipeline import I2VGenXLPipeline, _StepEngine  # noqa: E402
from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
pipe = I2VGenXLPipeline.from_pretrained("ali-vilab/i2vgen-xl", torch_dtype=torch.float16, variant="fp16", random_init_seed=0)
pipe.to(dev)
lat, ehs, ie, il_all = bench.synthetic_clip(dev, 8888)
inv, fwd = DDIMInverseScheduler(), DDIMScheduler()
inv.set_timesteps(50)
fwd.set_timesteps(50)
ts_inv, ts_pnp = [int(t) for t in inv.timesteps], [int(t) for t in fwd.timesteps]
pnp_utils.register_conv_injection(pipe, fwd.timesteps)
pnp_utils.register_spatial_attention_pnp(pipe, fwd.timesteps)
pnp_utils.register_temp_attention_pnp(pipe, fwd.timesteps)
s_inv, s_pnp = lat.clone(), lat.repeat(3, 1, 1, 1, 1).contiguous()
cond1 = dict(encoder_hidden_states=ehs[:1].contiguous(), fps=torch.tensor([8], device=dev), image_latents=il_all[:1].contiguous(),
             image_embeddings=ie[:1].contiguous())
cond3 = dict(encoder_hidden_states=ehs, fps=torch.tensor([8, 8, 8], device=dev), image_latents=il_all, image_embeddings=ie)
pnp_utils.clear_time(pipe)
e_inv = _StepEngine(pipe, s_inv, cond1, b_unc=-1, b_cond=0, guidance=1.0, dup_slots=[])
e_pnp = _StepEngine(pipe, s_pnp, cond3, b_unc=1, b_cond=2, guidance=9.0, dup_slots=[1], shared_stem=True)
tt_inv = torch.tensor(ts_inv, dtype=torch.float32, device=dev)[:, None].contiguous()
tt_pnp = torch.tensor(ts_pnp, dtype=torch.float32, device=dev)[:, None].expand(-1, 3).contiguous()
cf_inv, cf_pnp = inv.coefficient_table(ts_inv, dev), fwd.coefficient_table(ts_pnp, dev)
torch.cuda.synchronize()
print("STEP_TRAFFIC_BEGIN", flush=True)
for j in range(pairs):
    pnp_utils.clear_time(pipe)
    e_inv.step(tt_inv[j], cf_inv[j], key=("inv",))
    pnp_utils.register_time(pipe, ts_pnp[j])
    e_pnp.step(tt_pnp[j], cf_pnp[j], key=("pnp",) + pnp_utils.injection_state(pipe))
torch.cuda.synchronize()
print(f"STEP_TRAFFIC_PAIRS {pairs}", flush=True)
