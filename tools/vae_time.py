"""Where one clip's pre / post time goes around the loops: host preparation, H2D, VAE encode, VAE decode, frames -> PIL.  python tools/vae_time.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from PIL import Image
from anyv2v_amd.pipeline import I2VGenXLPipeline
from anyv2v_amd.encoders import _center_crop_wide, _pil_batch_to_device
torch.set_grad_enabled(False)
dev = torch.device("cuda")
pipe = I2VGenXLPipeline.from_pretrained("ali-vilab/i2vgen-xl", torch_dtype=torch.float16, variant="fp16", random_init_seed=0)
pipe.to(dev)
from anyv2v_amd.encoders import attach_native_vae
attach_native_vae(pipe, random_init_seed=0)
fr = [Image.fromarray(np.random.randint(0, 255, (512, 512, 3), dtype=np.uint8)) for _ in range(16)]
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = T(); lat = pipe.encode_vae_video(fr, dev, height=512, width=512); t1 = T()
    crops = [_center_crop_wide(f, (512, 512)) for f in fr]; t2 = T()
    xd = _pil_batch_to_device(crops, dev); t3 = T()
    mean, logvar = pipe.vae.model.encode_moments(xd); t4 = T()
    vid = pipe.decode_latents(lat, decode_chunk_size=1); t5 = T()
    pil = pipe.vae.to_pil(vid); t6 = T()
    vid4 = pipe.vae.decode_video(lat / 1.0, decode_chunk_size=16); t7 = T()
    print(f"rep {rep}: encode_vae_video {1e3*(t1-t0):.1f} ms | crop {1e3*(t2-t1):.1f} | uint8 H2D + table {1e3*(t3-t2):.1f} | encode_moments {1e3*(t4-t3):.1f} | decode_latents(chunk 1) {1e3*(t5-t4):.1f} | to_pil {1e3*(t6-t5):.1f} | decode(chunk 16) {1e3*(t7-t6):.1f}", flush=True)
