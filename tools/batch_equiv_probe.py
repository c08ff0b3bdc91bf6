"""Where does a [negative, editing] forward (replayed source features, batch hint 3/2) stop being bit-equal to branches 1, 2 of the
three-branch forward?  Full model, 16 f x 512^2, every hook site injected: the output of every ResNet / temporal-conv / transformer
module of both forwards is compared in network order.  gpurun_out/batch_equiv_probe.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from anyv2v_amd import ops, pnp_utils  # noqa: E402
from anyv2v_amd.pipeline import I2VGenXLPipeline  # noqa: E402
from anyv2v_amd.unet import ResnetBlock2D, TemporalConvLayer, Transformer2DModel, TransformerTemporalModel, BasicTransformerBlock, Attention, FeedForward  # noqa: E402

dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
pipe = I2VGenXLPipeline.from_pretrained("ali-vilab/i2vgen-xl", torch_dtype=torch.float16, variant="fp16", random_init_seed=0)
pipe.to(dev)
unet = pipe.unet
lat, ehs, ie, il = bench.synthetic_clip(dev, 8888)
smp = torch.cat([lat, lat * 0.9, lat * 0.9]).contiguous()
il = il.clone()
il[2] = il[1]
kw = dict(fps=torch.tensor([8, 8, 8], device=dev), image_latents=il, image_embeddings=ie, encoder_hidden_states=ehs)
ts = [981 - 20 * i for i in range(50)]
pnp_utils.register_conv_injection(pipe, ts)
pnp_utils.register_spatial_attention_pnp(pipe, ts)
pnp_utils.register_temp_attention_pnp(pipe, ts)
pnp_utils.register_time(pipe, 981)
sites = pnp_utils.injection_sites(pipe)
full = 16 * 64 * 64
bufs = {n: torch.zeros((full // {"1": 16, "2": 4, "3": 1}[n.split(".up")[1][0]], c), dtype=torch.float16, device=dev) for n, _o, c in sites}
rec = {}
order = []
kinds = (ResnetBlock2D, TemporalConvLayer, Transformer2DModel, TransformerTemporalModel, BasicTransformerBlock, Attention, FeedForward)
for name, m in unet.named_modules():
    if isinstance(m, kinds):
        orig = m.run

        def wrapped(*a, _o=orig, _n=name, **k):
            y = _o(*a, **k)
            t = y[0] if isinstance(y, tuple) else y
            rec.setdefault(cur[0], {})[_n] = t.clone()
            if cur[0] == "b3":
                order.append(_n)
            return y
        m.run = wrapped
cur = ["b3"]
for n, o, _c in sites:
    o.src_io = ("record", bufs[n])
def fwd(x, k, hint):
    """as a step engine runs it: shared stem on ([.., negative, editing] share latent and image latents), optional batch hint"""
    unet.forward_tokens(x, 981, k["fps"], k["image_latents"], k["image_embeddings"], k["encoder_hidden_states"])   # builds the clip context
    unet._ctx.shared_stem = True
    unet._ctx.batch_hint = hint
    if hint is None:
        return unet(x, 981, **k)[0]
    with ops.batch_hint(*hint):
        return unet(x, 981, **k)[0]


v3 = fwd(smp, kw, None)
cur = ["b2"]
for n, o, _c in sites:
    o.src_io = ("replay", bufs[n])
v2 = fwd(smp[1:].contiguous(), {k_: v_[1:].contiguous() for k_, v_ in kw.items()}, (3, 2))
lines = [f"final v-prediction: max |diff| {float((v2.float() - v3[1:].float()).abs().max()):.3e}"]
first = None
for n in order:
    a, b = rec["b3"][n], rec["b2"].get(n)
    if b is None:
        continue
    if a.shape[0] == b.shape[0] * 3 // 2:
        a = a[a.shape[0] // 3:]
    elif a.shape[0] == b.shape[0] * 2:      # shared stem: [source, shared] vs [shared]
        a = a[a.shape[0] // 2:]
    if a.shape != b.shape:
        lines.append(f"{n}: shapes {tuple(rec['b3'][n].shape)} vs {tuple(b.shape)} (not compared)")
        continue
    d = float((a.float() - b.float()).abs().max())
    if d > 0 and first is None:
        first = n
    if d > 0 or first is None:
        lines.append(f"{'DIFF ' if d > 0 else 'equal'} {n}: max |diff| {d:.3e}  (rows {b.shape[0]})")
    if first is not None and len([l for l in lines if l.startswith('DIFF')]) > 12:
        break
lines.append(f"first differing module: {first}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "batch_equiv_probe.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-40:]))
