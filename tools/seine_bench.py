"""SEINE at the released model's width (Stable-Diffusion-1.4 layout, 8 heads of 40 / 80 / 160 channels, 9 input channels) on the GPU:
time of one UNet forward at B = 1 (inversion step) and B = 3 (PnP edit step, all four hook families on) for 16 frames at 320 x 512 and
256 x 256, random weights, eager launches.  `python tools/seine_bench.py [h w]` -> one JSON line."""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyv2v_amd import seine as sn  # noqa: E402
from anyv2v_amd.consisti2v_pipeline import init_random_weights_  # noqa: E402
from anyv2v_amd.seine_pipeline import SEINE_UNET_CONFIG  # noqa: E402


def main():
    h = int(sys.argv[1]) if len(sys.argv) > 2 else 320
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    dev = torch.device("cuda:0")
    unet = init_random_weights_(sn.UNet3DConditionModel(**SEINE_UNET_CONFIG), 0).to(dev)
    n_params = sum(p.numel() for p in unet.parameters())
    model = types.SimpleNamespace(unet=unet)
    out = {}
    for B, hooks in ((1, False), (3, True)):
        x = torch.randn(B, 9, 16, h // 8, w // 8, device=dev).half()
        ehs = torch.randn(B, 77, 768, device=dev).half()
        if hooks:
            for reg in (sn.register_conv_injection, sn.register_spatial_attention_pnp, sn.register_cross_attention_pnp, sn.register_temp_attention_pnp):
                reg(model, [981])
            sn.register_time(model, 981)
        y = unet(x, 981, encoder_hidden_states=ehs).sample
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            y = unet(x, 981, encoder_hidden_states=ehs).sample
        torch.cuda.synchronize()
        out[f"B{B}_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
        out[f"B{B}_finite"] = bool(torch.isfinite(y.float()).all())
    print(json.dumps(dict(what="SEINE UNet forward at the released width, eager", height=h, width=w, frames=16, unet_params_M=round(n_params / 1e6, 1), **out)))


if __name__ == "__main__":
    main()
