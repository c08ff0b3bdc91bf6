"""ConsistI2V at the released model's width on the GPU: per-step time of the inversion (B = 1) and PnP-edit (B = 3, hooks on) UNet
forward + guided step at 16 frames x 256^2 / 512^2 (the reference's configs/pipeline_256, pipeline_512), random weights, eager
launches.  `python tools/consisti2v_bench.py [256|512] [steps]` -> one JSON line.  Not the headline bench (bench.py is I2VGen-XL)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyv2v_amd import consisti2v as c2  # noqa: E402
from anyv2v_amd.consisti2v_pipeline import ConditionalVideoEditingPipeline  # noqa: E402
from anyv2v_amd.schedulers import CONSISTI2V_SCHEDULER_CONFIG, DDIMInverseScheduler, DDIMScheduler  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda:0")
    pipe = ConditionalVideoEditingPipeline.from_pretrained("TIGER-Lab/ConsistI2V", random_init_seed=0).to(dev)
    n_params = sum(p.numel() for p in pipe.unet.parameters())
    from PIL import Image
    import numpy as np
    rng = np.random.RandomState(0)
    frames = [Image.fromarray(rng.randint(0, 255, (size, size, 3), dtype=np.uint8)) for _ in range(16)]
    inv = DDIMInverseScheduler(**CONSISTI2V_SCHEDULER_CONFIG)
    pipe.register_modules(scheduler=inv)
    lat0 = pipe.encode_vae_video(frames, dev, height=size, width=size)

    def timed(fn):
        fn()                      # warm-up (packing, index tensors)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t0)
    traj, t_inv = timed(lambda: pipe.invert(prompt="", first_frame_paths=frames[0], height=size, width=size, video_length=16,
                                            num_inference_steps=n, guidance_scale_txt=1.0, guidance_scale_img=1.0, negative_prompt="",
                                            frame_stride=3, latents=lat0, return_trajectory=True))
    fwd = DDIMScheduler(**CONSISTI2V_SCHEDULER_CONFIG)
    fwd.set_timesteps(n)
    pipe.register_modules(scheduler=fwd)
    ts = fwd.timesteps
    c2.register_conv_injection(pipe, ts)
    c2.register_spatial_attention_pnp(pipe, ts)
    c2.register_temp_attention_pnp(pipe, ts)
    T = int(ts[0])
    ed, t_ed = timed(lambda: pipe.sample_with_pnp(prompt="a robot", first_frame_paths=frames[1], height=size, width=size, video_length=16,
                                                  num_inference_steps=n, guidance_scale_txt=35.0, guidance_scale_img=1.0, negative_prompt="blurry",
                                                  frame_stride=3, latents=traj[T].clone(), ddim_init_latents_t_idx=0,
                                                  ddim_inv_latents_path=traj, ddim_inv_prompt="", ddim_inv_1st_frame_path=frames[0],
                                                  output_type="latent").videos)
    print(json.dumps(dict(what="ConsistI2V full width", size=size, frames=16, steps=n, unet_params_M=round(n_params / 1e6, 1),
                          inversion_ms_per_step=round(1e3 * t_inv / n, 2), pnp_edit_ms_per_step=round(1e3 * t_ed / n, 2),
                          hip_graphs=os.environ.get("ANYV2V_CONSISTI2V_GRAPHS", "0") == "1", finite=bool(torch.isfinite(ed.float()).all()), edit_absmax=float(ed.float().abs().max()))))


if __name__ == "__main__":
    main()
