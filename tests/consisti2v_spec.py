"""Shared definition of the ConsistI2V decoder-hook fixture (tests/golden/consisti2v_decoder_hooks.pt): block shapes, weights and
inputs are re-derived from names and seeds on both sides -- ``tests/golden/make_golden.py --consisti2v`` (the REFERENCE's own
``VideoLDMCrossAttnUpBlock`` + ``consisti2v/pnp_utils.py``, CPU fp32) and the tests (native blocks, op emulation on the CPU, HIP
kernels on the GPU) -- so the fixture holds outputs only."""
import types
import zlib

import torch

B, FR, H, W = 3, 4, 3, 4          # [source, negative, editing] x 4 frames, 3 x 4 latent pixels (not square on purpose)
TEMB, CROSS, TOKENS, GROUPS = 64, 48, 7, 8
N_STEPS = 50
PNP = dict(pnp_f_t=0.2, pnp_spatial_attn_t=0.5, pnp_temp_attn_t=0.8)
TS_CASES = (981, 301)             # every site injecting / temporal attention only
WEIGHT_SEED, INPUT_SEED = 4321, 8888

# stand-ins for unet.up_blocks[1..3] (three layers each, as the reference builds them: layers_per_block + 1)
BLOCKS = {
    1: dict(in_channels=64, out_channels=128, prev_output_channel=128, num_attention_heads=2, n_temp_heads=4, add_upsample=True),
    2: dict(in_channels=64, out_channels=64, prev_output_channel=128, num_attention_heads=1, n_temp_heads=4, add_upsample=True),
    3: dict(in_channels=64, out_channels=64, prev_output_channel=64, num_attention_heads=1, n_temp_heads=1, add_upsample=False),
}


def block_kwargs(i):
    return dict(temb_channels=TEMB, num_layers=3, resnet_eps=1e-5, resnet_groups=GROUPS, cross_attention_dim=CROSS,
                use_linear_projection=True, use_temporal=True, augment_temporal_attention=True, n_frames=FR,
                first_frame_condition_mode="concat", rotary_emb=True, **BLOCKS[i])


def fill_weights(module, seed=WEIGHT_SEED):
    """Deterministic parameters by NAME (both module trees have the reference's state-dict keys)."""
    sd = module.state_dict()
    new = {}
    for name in sorted(sd):
        v = sd[name]
        g = torch.Generator().manual_seed(seed + zlib.crc32(name.encode()) % 100000)
        if name.endswith("freqs"):
            new[name] = v.clone()
        elif name.endswith("alpha"):
            new[name] = torch.rand(1, generator=g) * 0.6 + 0.2
        elif v.dim() >= 2:
            fan_in = v[0].numel()
            new[name] = torch.randn(v.shape, generator=g) * (1.0 / fan_in ** 0.5)
        elif "norm" in name and name.endswith("weight"):
            new[name] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            new[name] = 0.1 * torch.randn(v.shape, generator=g)
    module.load_state_dict(new)
    return module


def block_inputs(i, seed=INPUT_SEED):
    k = BLOCKS[i]
    g = torch.Generator().manual_seed(seed + i)
    N = B * FR
    cout, cin, prev = k["out_channels"], k["in_channels"], k["prev_output_channel"]
    r = lambda *s: torch.randn(*s, generator=g).half().float()     # the tests feed fp16 tensors: generate fp16-representable values
    x = r(N, prev, H, W)
    skips = (r(N, cin, H, W), r(N, cout, H, W), r(N, cout, H, W))  # consumed from the end
    temb = r(B, TEMB).repeat_interleave(FR, 0)
    ehs = r(B, TOKENS, CROSS).repeat_interleave(FR, 0)
    return x, skips, temb, ehs


def schedules():
    ts = torch.arange(N_STEPS).flip(0) * (1000 // N_STEPS) + 1
    return (ts[: int(N_STEPS * PNP["pnp_f_t"])], ts[: int(N_STEPS * PNP["pnp_spatial_attn_t"])],
            ts[: int(N_STEPS * PNP["pnp_temp_attn_t"])])


def stub_model(blocks):
    """``model.unet.up_blocks[i]`` as the hook functions index it (``consisti2v/pnp_utils.py:20-28``); index 0 holds no hook site."""
    return types.SimpleNamespace(unet=types.SimpleNamespace(up_blocks=[None, blocks[1], blocks[2], blocks[3]]))


def run_cases(blocks, pnp_module, call, log=None):
    """Outputs of every block: un-hooked, then with the hook family registered at each timestep of TS_CASES."""
    out = {}
    for i, blk in blocks.items():
        out[f"block{i}_nohook"] = call(blk, *block_inputs(i))
    model = stub_model(blocks)
    conv_s, spa_s, tmp_s = schedules()
    pnp_module.register_conv_injection(model, conv_s)
    pnp_module.register_spatial_attention_pnp(model, spa_s)
    pnp_module.register_temp_attention_pnp(model, tmp_s)
    for t in TS_CASES + (101,):
        pnp_module.register_time(model, t)
        for i, blk in blocks.items():
            out[f"block{i}_hook_t{t}"] = call(blk, *block_inputs(i))
    return out


# ------------------------------------------------------------------------------------------------- the whole UNet (consisti2v_unet.pt)
# VideoLDMUNet3DConditionModel with the released model's options at toy width: 4 levels (the hooks index up_blocks[1..3]), 2 layers
# per block, first-frame conditioning by concatenation, augmented rotary temporal attention, frame-stride conditioning.
UNET_CFG = dict(sample_size=8, in_channels=4, out_channels=4, block_out_channels=(32, 64, 64, 64), layers_per_block=2, norm_num_groups=8,
                cross_attention_dim=48, attention_head_dim=(2, 4, 4, 4), use_linear_projection=True, use_temporal=True, n_frames=4,
                n_temp_heads=2, first_frame_condition_mode="concat", augment_temporal_attention=True, temp_pos_embedding="rotary",
                use_frame_stride_condition=True)
UNET_H, UNET_W, UNET_T, UNET_STRIDE = 8, 16, 981, 3


def unet_inputs(seed=INPUT_SEED):
    """[source, negative, editing] x (n_frames - 1) noisy frames, the clean first-frame latent, text tokens."""
    g = torch.Generator().manual_seed(seed + 77)
    r = lambda *s: torch.randn(*s, generator=g).half().float()
    nf = UNET_CFG["n_frames"] - 1
    sample = r(B, UNET_CFG["in_channels"], nf, UNET_H, UNET_W)
    first = r(1, UNET_CFG["in_channels"], 1, UNET_H, UNET_W).repeat(B, 1, 1, 1, 1)   # the same clip's first frame in every branch
    ehs = r(B, TOKENS, UNET_CFG["cross_attention_dim"])
    return sample, first, ehs


def run_unet_cases(unet, pnp_module, call):
    """``call(unet, sample, t, ehs, first, frame_stride)`` -> prediction; un-hooked, then hooked at each timestep of TS_CASES (+ 101)."""
    out = {}
    sample, first, ehs = unet_inputs()
    out["unet_nohook"] = call(unet, sample, UNET_T, ehs, first, UNET_STRIDE)
    out["unet_nohook_t101"] = call(unet, sample, 101, ehs, first, UNET_STRIDE)
    model = types.SimpleNamespace(unet=unet)
    conv_s, spa_s, tmp_s = schedules()
    pnp_module.register_conv_injection(model, conv_s)
    pnp_module.register_spatial_attention_pnp(model, spa_s)
    pnp_module.register_temp_attention_pnp(model, tmp_s)
    for t in TS_CASES + (101,):
        pnp_module.register_time(model, t)
        out[f"unet_hook_t{t}"] = call(unet, sample, t, ehs, first, UNET_STRIDE)
    return out
