"""Shared definition of the SEINE decoder-hook fixture (tests/golden/seine_decoder_hooks.pt): see consisti2v_spec.py -- weights and
inputs are re-derived from names and seeds by the generator (the REFERENCE's own ``CrossAttnUpBlock3D`` + ``seine/pnp_utils.py``,
``tests/golden/make_golden.py --seine``) and by the tests (native blocks); the fixture holds outputs only."""
import types

import torch
from torch import nn

from consisti2v_spec import B, CROSS, FR, GROUPS, H, INPUT_SEED, N_STEPS, TEMB, TOKENS, W, fill_weights  # noqa: F401 (same geometry)

PNP = dict(pnp_f_t=0.2, pnp_spatial_attn_t=0.4, pnp_cross_attn_t=0.6, pnp_temp_attn_t=0.8)
TS_CASES = (981, 501, 301)        # every hook injecting / cross + temporal / temporal only
WEIGHT_SEED = 9753

BLOCKS = {   # stand-ins for unet.up_blocks[1..3]; attn_num_head_channels = number of heads (seine/models/unet_blocks.py:498-500)
    1: dict(in_channels=64, out_channels=128, prev_output_channel=128, attn_num_head_channels=2, add_upsample=True),
    2: dict(in_channels=64, out_channels=64, prev_output_channel=128, attn_num_head_channels=2, add_upsample=True),
    3: dict(in_channels=64, out_channels=64, prev_output_channel=64, attn_num_head_channels=1, add_upsample=False),
}


def block_kwargs(i):
    return dict(temb_channels=TEMB, num_layers=3, resnet_eps=1e-5, resnet_groups=GROUPS, cross_attention_dim=CROSS,
                use_linear_projection=(i == 2), use_first_frame=False, use_relative_position=False, **BLOCKS[i])


def block_inputs(i, seed=INPUT_SEED):
    k = BLOCKS[i]
    g = torch.Generator().manual_seed(seed + 100 + i)
    cout, cin, prev = k["out_channels"], k["in_channels"], k["prev_output_channel"]
    r = lambda *s: torch.randn(*s, generator=g).half().float()
    x = r(B, prev, FR, H, W)
    skips = (r(B, cin, FR, H, W), r(B, cout, FR, H, W), r(B, cout, FR, H, W))
    return x, skips, r(B, TEMB), r(B, TOKENS, CROSS)


def schedules():
    ts = torch.arange(N_STEPS).flip(0) * (1000 // N_STEPS) + 1
    return {k: ts[: int(N_STEPS * v)] for k, v in PNP.items()}


class StubUNet(nn.Module):
    """``model.unet`` as ``seine/pnp_utils.py`` walks it: ``up_blocks[1..3]`` are the blocks under test; ``down_blocks`` / ``mid_block``
    only receive ``register_time``'s attribute writes (``:133-147``) and hold plain namespaces."""

    def __init__(self, blocks):
        super().__init__()
        self.up_blocks = nn.ModuleList([nn.Identity(), blocks[1], blocks[2], blocks[3]])
        site = lambda: types.SimpleNamespace(transformer_blocks=[types.SimpleNamespace(
            attn1=types.SimpleNamespace(), attn2=types.SimpleNamespace(), attn_temp=types.SimpleNamespace())])
        self.__dict__["down_blocks"] = [types.SimpleNamespace(attentions=[site(), site()]) for _ in range(3)]
        self.__dict__["mid_block"] = types.SimpleNamespace(attentions=[site()])


def run_cases(blocks, pnp_module, call):
    out = {}
    for i, blk in blocks.items():
        out[f"block{i}_nohook"] = call(blk, *block_inputs(i))
    model = types.SimpleNamespace(unet=StubUNet(blocks))
    s = schedules()
    pnp_module.register_conv_injection(model, s["pnp_f_t"])
    pnp_module.register_spatial_attention_pnp(model, s["pnp_spatial_attn_t"])
    pnp_module.register_cross_attention_pnp(model, s["pnp_cross_attn_t"])
    pnp_module.register_temp_attention_pnp(model, s["pnp_temp_attn_t"])
    for t in TS_CASES + (101,):
        pnp_module.register_time(model, t)
        for i, blk in blocks.items():
            out[f"block{i}_hook_t{t}"] = call(blk, *block_inputs(i))
    return out


# ------------------------------------------------------------------------------------------------- the whole UNet (seine_unet.pt)
# UNet3DConditionModel with the released model's options at toy width: 4 levels, 2 layers per block, 9 input channels
# (latents | mask | masked-video latents), 1 x 1-conv projections, head_dim >= 32 everywhere (RotaryEmbedding(32) per head).
UNET_CFG = dict(sample_size=8, in_channels=9, out_channels=4, block_out_channels=(64, 64, 128, 128), layers_per_block=2, norm_num_groups=8,
                cross_attention_dim=CROSS, attention_head_dim=(2, 1, 2, 4), use_linear_projection=False)
UNET_H, UNET_W, UNET_F = 8, 16, 4


def unet_inputs(seed=INPUT_SEED):
    g = torch.Generator().manual_seed(seed + 277)
    r = lambda *s: torch.randn(*s, generator=g).half().float()
    return r(B, 9, UNET_F, UNET_H, UNET_W), r(B, TOKENS, CROSS)


def run_unet_cases(unet, pnp_module, call):
    """``call(unet, sample, t, ehs)`` -> prediction; un-hooked (t = 981, 101), then with the FOUR hook families at each TS_CASES (+ 101)."""
    out = {}
    sample, ehs = unet_inputs()
    out["unet_nohook"] = call(unet, sample, 981, ehs)
    out["unet_nohook_t101"] = call(unet, sample, 101, ehs)
    model = types.SimpleNamespace(unet=unet)
    s = schedules()
    pnp_module.register_conv_injection(model, s["pnp_f_t"])
    pnp_module.register_spatial_attention_pnp(model, s["pnp_spatial_attn_t"])
    pnp_module.register_cross_attention_pnp(model, s["pnp_cross_attn_t"])
    pnp_module.register_temp_attention_pnp(model, s["pnp_temp_attn_t"])
    for t in TS_CASES + (101,):
        pnp_module.register_time(model, t)
        out[f"unet_hook_t{t}"] = call(unet, sample, t, ehs)
    return out
