"""Several clips inverted in one batch (`pipe.invert_clips`) vs one by one (`pipe.invert`): seconds per clip for a 50-step inversion at
16 f x 512^2, full-width I2VGen-XL UNet with random weights, synthetic VAE / CLIP stand-ins, HIP graphs warm.
`python tools/batched_inversion_probe.py [B ...]` -> one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyv2v_amd.encoders import attach_synthetic_encoders  # noqa: E402
from anyv2v_amd.pipeline import I2VGenXLPipeline  # noqa: E402
from anyv2v_amd.schedulers import DDIMInverseScheduler  # noqa: E402


def main():
    bs = [int(a) for a in sys.argv[1:]] or [2, 4]
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    pipe = I2VGenXLPipeline.from_pretrained("ali-vilab/i2vgen-xl", torch_dtype=torch.float16, variant="fp16", random_init_seed=0).to(dev)
    attach_synthetic_encoders(pipe)
    pipe.scheduler = DDIMInverseScheduler()
    rng = np.random.RandomState(0)
    n = max(bs)
    clips = []
    for k in range(n):
        frames = [Image.fromarray((rng.rand(512, 512, 3) * 255).astype("uint8")) for _ in range(16)]
        clips.append(dict(prompt="", image=frames[0], latents=pipe.encode_vae_video(frames, dev, height=512, width=512)))
    kw = dict(height=512, width=512, num_frames=16, num_inference_steps=50, target_fps=8)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0
    single, t1 = timed(lambda: [pipe.invert(prompt=c["prompt"], image=c["image"], latents=c["latents"], guidance_scale=1.0, return_trajectory=True, **kw)
                                for c in clips])
    res = dict(what="50-step inversion at 16 f x 512^2, seconds per clip", one_by_one=round(t1 / n, 3))
    T = max(single[0].keys())
    for b in bs:
        groups = [clips[i:i + b] for i in range(0, n, b)]
        out, tb = timed(lambda: [pipe.invert_clips(g, **kw) for g in groups])
        res[f"batch_{b}"] = round(tb / n, 3)
        a, r = out[0][0][T].float(), single[0][T].float()
        res[f"batch_{b}_vs_single_max_rel"] = round(float((a - r).abs().max() / r.abs().max()), 5)
    res["speedup"] = {f"batch_{b}": round(res["one_by_one"] / res[f"batch_{b}"], 3) for b in bs}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
