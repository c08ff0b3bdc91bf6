"""Generate tests/golden/*.pt by running the REFERENCE's own code (only works where /root/reference exists).

    python tests/golden/make_golden.py

1. ``pnp_hooks_mini.pt``: the reference's ``i2vgen-xl/pnp_utils.py`` (imported verbatim via
   ``oracle.ref_stubs``) registered on the mini oracle UNet; v-predictions for a 3-way batch at
   timesteps that are on all / some / none of the injection schedules, plus the un-hooked output.
   ``pnp_hooks_full_config1.pt`` (``--full``, ~1 min of CPU): the same at full width -- the 1.42 B-parameter oracle at
   BASELINE config 1 (3 x 8 f x 256^2, weights seed 1234, inputs seed 8888 = ``tests/gpu_checks.config1_inputs``), the
   reference's hooks registered on all 17 sites; v-predictions at t=981 (every site injecting) and t=301 (temporal only).
2. ``inverse_scheduler.pt``: the reference's vendored ``consisti2v/ddim_inverse_scheduler.py``
   constructed with the config logged at ``i2vgen-xl/demo.ipynb:1208-1226``: alphas_cumprod table,
   timesteps for n=50/500, and inverse steps on seeded tensors.
"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_stubs  # noqa: E402
from oracle.unet_oracle import UNetConfig, build_oracle, random_state_dict  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

MINI_SEED = 1234
INPUT_SEED = 8888  # the reference's seed (configs/group_pnp_edit/template.yaml:4)
B, FR, HW = 3, 4, 8
N_STEPS = 50
PNP = dict(pnp_f_t=0.2, pnp_spatial_attn_t=0.5, pnp_temp_attn_t=0.8)  # -> 10 / 25 / 40 steps


def mini_inputs(cfg):
    g = torch.Generator().manual_seed(INPUT_SEED)
    sample = torch.randn(B, 4, FR, HW, HW, generator=g)
    il = torch.randn(B, 4, FR, HW, HW, generator=g)
    for i in range(1, FR):  # frame-position planes, pipeline_i2vgen_xl.py:548-554
        il[:, :, i] = i / (FR - 1)
    ehs = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    ie = torch.randn(B, 1, cfg.cross_attention_dim, generator=g)
    return dict(sample=sample, image_latents=il, encoder_hidden_states=ehs, image_embeddings=ie,
                fps=torch.tensor([8] * B))


def gen_hooks():
    ref = ref_stubs.load_reference_pnp_utils()
    cfg = UNetConfig.mini()
    unet = build_oracle(cfg, random_state_dict(cfg, MINI_SEED), dtype=torch.float32)
    inp = mini_inputs(cfg)
    kw = dict(fps=inp["fps"], image_latents=inp["image_latents"], image_embeddings=inp["image_embeddings"],
              encoder_hidden_states=inp["encoder_hidden_states"])
    out = {"inputs": inp, "mini_seed": MINI_SEED, "pnp": PNP, "n_steps": N_STEPS}
    with torch.no_grad():
        out["v_nohook_t981"] = unet(inp["sample"], 981, **kw)[0].clone()
        pipe = types.SimpleNamespace(unet=unet)
        ts = torch.arange(N_STEPS).flip(0) * (1000 // N_STEPS) + 1  # 981..1, demo.ipynb:1201-1204
        ref.register_conv_injection(pipe, ts[: int(N_STEPS * PNP["pnp_f_t"])])
        ref.register_spatial_attention_pnp(pipe, ts[: int(N_STEPS * PNP["pnp_spatial_attn_t"])])
        ref.register_temp_attention_pnp(pipe, ts[: int(N_STEPS * PNP["pnp_temp_attn_t"])])
        for t in (981, 701, 301, 101):  # on all three / spatial+temp / temp only / none
            ref.register_time(pipe, t)
            out[f"v_hook_t{t}"] = unet(inp["sample"], t, **kw)[0].clone()
    torch.save(out, os.path.join(HERE, "pnp_hooks_mini.pt"))
    for k, v in out.items():
        if torch.is_tensor(v):
            print(k, tuple(v.shape), float(v.abs().max()))


def gen_hooks_full():
    """Full-width fixture.  Only outputs are stored (2 x 393 KB fp32): weights and inputs are
    re-derived from their seeds by the tests."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_checks as gc
    ref = ref_stubs.load_reference_pnp_utils()
    cfg = UNetConfig.i2vgen_xl()
    unet = build_oracle(cfg, random_state_dict(cfg, MINI_SEED), dtype=torch.float32)
    inp = gc.config1_inputs(cfg, 3, 8, 32, seed=INPUT_SEED)
    inp = {k: (v.half().float() if v.is_floating_point() else v) for k, v in inp.items()}   # the tests feed fp16-rounded inputs
    kw = dict(fps=inp["fps"], image_latents=inp["image_latents"], image_embeddings=inp["image_embeddings"],
              encoder_hidden_states=inp["encoder_hidden_states"])
    out = {"weights_seed": MINI_SEED, "input_seed": INPUT_SEED, "pnp": PNP, "n_steps": N_STEPS, "shape": tuple(inp["sample"].shape)}
    with torch.no_grad():
        pipe = types.SimpleNamespace(unet=unet)
        ts = torch.arange(N_STEPS).flip(0) * (1000 // N_STEPS) + 1
        ref.register_conv_injection(pipe, ts[: int(N_STEPS * PNP["pnp_f_t"])])
        ref.register_spatial_attention_pnp(pipe, ts[: int(N_STEPS * PNP["pnp_spatial_attn_t"])])
        ref.register_temp_attention_pnp(pipe, ts[: int(N_STEPS * PNP["pnp_temp_attn_t"])])
        for t in (981, 301):
            ref.register_time(pipe, t)
            out[f"v_hook_t{t}"] = unet(inp["sample"], t, **kw)[0].clone()
            print(f"v_hook_t{t}", tuple(out[f"v_hook_t{t}"].shape), float(out[f"v_hook_t{t}"].abs().max()))
    torch.save(out, os.path.join(HERE, "pnp_hooks_full_config1.pt"))


def gen_scheduler():
    mod = ref_stubs.load_reference_inverse_scheduler()
    cfgd = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="squaredcos_cap_v2",
                clip_sample=False, set_alpha_to_one=True, steps_offset=1, prediction_type="v_prediction",
                timestep_spacing="leading", rescale_betas_zero_snr=True)  # demo.ipynb:1208-1226
    s = mod.DDIMInverseScheduler(**cfgd)
    out = {"alphas_cumprod": s.alphas_cumprod.clone(), "config": cfgd}
    g = torch.Generator().manual_seed(INPUT_SEED)
    x = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    v = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    out["x"], out["v"] = x, v
    for n in (50, 500):
        s.set_timesteps(n)
        out[f"timesteps_{n}"] = s.timesteps.clone()
        for t in (1, int(s.timesteps[1]), int(s.timesteps[n // 2]), int(s.timesteps[-1])):
            out[f"inv_step_n{n}_t{t}"] = s.step(v, t, x).prev_sample.clone()
    torch.save(out, os.path.join(HERE, "inverse_scheduler.pt"))
    print("alphas_cumprod[0,500,998,999] =", [float(out["alphas_cumprod"][i]) for i in (0, 500, 998, 999)])
    print("timesteps_50[:5] =", out["timesteps_50"][:5].tolist(), " timesteps_500[-3:] =", out["timesteps_500"][-3:].tolist())


if __name__ == "__main__":
    assert ref_stubs.reference_available(), "needs /root/reference"
    if "--full" in sys.argv:
        gen_hooks_full()
    else:
        gen_hooks()
        gen_scheduler()
