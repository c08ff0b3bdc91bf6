"""BASELINE config 5 (1 clip x 128 frames x 512x512) at full size: one inversion step (B=1) and one PnP edit step (B=3,
all injections on) of the full UNet -- does it run, is it finite, how long does it take, how much HBM does it need.
gpurun_out/config5_probe.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd import pnp_utils  # noqa: E402
from anyv2v_amd.pipeline import I2VGenXLPipeline, _StepEngine  # noqa: E402
from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler  # noqa: E402

FR, LAT = int(os.environ.get("FRAMES", "128")), 64
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
pipe = I2VGenXLPipeline.from_pretrained("ali-vilab/i2vgen-xl", torch_dtype=torch.float16, variant="fp16", random_init_seed=0)
pipe.to(dev)
g = torch.Generator().manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g).to(torch.float16).to(dev)
lat, ehs, ie, il = r(1, 4, FR, LAT, LAT), r(3, 77, 1024), r(3, 1, 1024), r(2, 4, FR, LAT, LAT)
ie[1].zero_()
for i in range(1, FR):
    il[:, :, i] = i / (FR - 1)
il_all = torch.stack([il[0], il[1], il[1]]).contiguous()
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
lines = []
for name, mk, tt, cf, key, reg in (
        ("inversion step (B=1)", lambda: _StepEngine(pipe, s_inv, cond1, b_unc=-1, b_cond=0, guidance=1.0, dup_slots=[]),
         torch.tensor(ts_inv, dtype=torch.float32, device=dev)[:, None].contiguous(), inv.coefficient_table(ts_inv, dev), ("inv",), None),
        ("PnP edit step (B=3, all injections)", lambda: _StepEngine(pipe, s_pnp, cond3, b_unc=1, b_cond=2, guidance=9.0, dup_slots=[1], shared_stem=True),
         torch.tensor(ts_pnp, dtype=torch.float32, device=dev)[:, None].expand(-1, 3).contiguous(), fwd.coefficient_table(ts_pnp, dev),
         ("pnp",), ts_pnp)):
    if reg is None:
        pnp_utils.clear_time(pipe)
    else:
        pnp_utils.register_time(pipe, reg[0])
    eng = mk()
    eng.step(tt[0], cf[0], key=key)   # warm-up + graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in (1, 2):
        eng.step(tt[j], cf[j], key=key)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    smp = s_inv if reg is None else s_pnp
    lines.append(f"{FR} frames x 512x512, {name}: {ms:8.1f} ms per step, finite={bool(torch.isfinite(smp.float()).all())}, "
                 f"peak HBM {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")
    print(lines[-1], flush=True)
    del eng
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "config5_probe.txt"), "w").write("\n".join(lines) + "\n")
