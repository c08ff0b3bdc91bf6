"""Native AutoencoderKL at the clip size of the bench (16 frames x 512x512): encode / decode wall time (GPU).
gpurun_out/vae_probe.txt        python tools/vae_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anyv2v_amd.vae import AutoencoderKL, init_random_weights_  # noqa: E402

dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
vae = init_random_weights_(AutoencoderKL(), 0).to(dev)
x = torch.rand(16, 3, 512, 512, device=dev) * 2 - 1
z = torch.randn(16, 4, 64, 64, device=dev)
lines = []
for name, fn in (("encode 16 x 3x512x512 -> posterior moments", lambda: vae.encode_moments(x)),
                 ("decode 16 x 4x64x64 -> 16 x 3x512x512 (one chunk)", lambda: vae.decode(z)),
                 ("decode, 4 frames per chunk", lambda: [vae.decode(z[i:i + 4]) for i in range(0, 16, 4)])):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    lines.append(f"{name}: {ms:8.1f} ms   (peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB)")
    print(lines[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "vae_probe.txt"), "w").write("\n".join(lines) + "\n")
