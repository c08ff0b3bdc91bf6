"""Host logic on CPU (no GPU): the native UNet / hooks / schedulers / pipeline loops driven through a TEST-ONLY
emulation of the C-ABI ops (tests/cpu_ops_emulation.py), compared with the oracle and the reference-generated
golden fixtures.  This validates token-layout wiring, weight packing, the per-clip conditioning cache, PnP
aliasing and dead-compute elimination, and the step engine -- everything except the HIP kernels themselves, which
the `-m gpu` tests cover through the real C ABI."""
import os

import numpy as np
import pytest
import torch

import cpu_ops_emulation as emu
import gpu_checks as gc


@pytest.fixture()
def cpu_ops(monkeypatch):
    emu.install(monkeypatch)
    monkeypatch.setattr(gc, "DEV", "cpu")
    monkeypatch.setenv("ANYV2V_NO_GRAPH", "1")
    yield


def _ok(results):
    bad = [f"{r['name']}: {r['err']:.3e} > {r['tol']:.1e}" for r in results if not r["ok"]]
    assert not bad, "\n".join(bad)


def test_state_dict_keys_match_diffusers_naming():
    from anyv2v_amd.unet import I2VGenXLUNet, I2VGenXLUNetConfig
    from oracle.unet_oracle import I2VGenXLUNetOracle, UNetConfig
    with torch.device("meta"):
        n, o = I2VGenXLUNet(I2VGenXLUNetConfig()), I2VGenXLUNetOracle(UNetConfig.i2vgen_xl())
    sn = {k: tuple(v.shape) for k, v in n.state_dict().items()}
    so_ = {k: tuple(v.shape) for k, v in o.state_dict().items()}
    assert sn == so_
    for k in ("down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight", "up_blocks.1.resnets.1.conv_shortcut.weight",
              "mid_block.temp_convs.0.conv4.3.weight", "transformer_in.transformer_blocks.0.ff.net.0.proj.bias",
              "image_latents_context_embedding.5.weight", "up_blocks.2.upsamplers.0.conv.bias"):
        assert k in sn
    # hook attribute paths (i2vgen-xl/pnp_utils.py:20-27,130,239,344)
    assert hasattr(n.up_blocks[1].resnets[1], "conv_shortcut")
    assert hasattr(n.up_blocks[3].temp_attentions[2].transformer_blocks[0].attn1, "processor")


def test_native_unet_vs_reference_generated_golden(cpu_ops):
    _ok(gc.check_unet_golden())


def test_reference_pnp_utils_drives_the_native_unet(cpu_ops):
    """Seams B1 / B2: the reference's OWN ``i2vgen-xl/pnp_utils.py`` (verbatim, via oracle.ref_stubs) registered on
    ``anyv2v_amd.unet.I2VGenXLUNet`` -- 16 foreign attention processors + the replaced ResNet ``forward`` -- reproduces
    the reference-generated fixture and the native hooks."""
    from oracle import ref_stubs
    if not ref_stubs.reference_available():
        pytest.skip("needs /root/reference")
    _ok(gc.check_foreign_hooks("reference"))


def test_oracle_hook_code_drives_the_native_unet(cpu_ops):
    _ok(gc.check_foreign_hooks("oracle"))


def test_native_unet_vs_oracle_with_and_without_pnp(cpu_ops):
    _ok(gc.check_unet_vs_oracle("mini", 3, 4, 8))
    _ok(gc.check_unet_vs_oracle("mini", 1, 8, 16, with_pnp=False))


def test_native_unet_vs_oracle_at_latent_sizes_that_are_not_multiples_of_8(cpu_ops):
    """The front ends take clips at their own size (``gradio_demo.py:129``, ``predict.py:153``): 512 x 288 pixels are 64 x 36 latents,
    and 36 -> 18 -> 9 -> 5 does not come back by doubling.  diffusers' UNet then hands every up block the size of the skip connections
    ahead (``forward_upsample_size``); oracle and native UNet both follow that rule (pinned to in-tree reference code through the
    sibling UNets: tests/test_seine.py, tests/test_consisti2v.py).  Rectangular, odd, and with the PnP hooks on."""
    _ok(gc.check_unet_vs_oracle("mini", 1, 4, (8, 16), with_pnp=False))
    _ok(gc.check_unet_vs_oracle("mini", 1, 2, (9, 10), with_pnp=False))
    _ok(gc.check_unet_vs_oracle("mini", 3, 2, (12, 9)))
    from anyv2v_amd.unet import upsample_tokens
    with pytest.raises(ValueError):
        upsample_tokens(None, torch.zeros(6, 4, dtype=torch.float16), 2, 3, (6, 6))


def test_pipeline_loops_vs_oracle(cpu_ops):
    _ok(gc.check_loops_mini())


def test_source_feature_cache_multi_edit_is_bit_equal(cpu_ops):
    """Several edits of one clip: record once, replay for the further edits -- bit-equal to the uncached run (CPU op emulation)."""
    _ok(gc.check_source_cache())


def test_step_engines_are_reused_across_clips_without_stale_state(cpu_ops):
    _ok(gc.check_engine_reuse_across_clips())


def test_kernel_check_references_are_self_consistent(cpu_ops):
    """The references used by the gpu kernel tests agree with the op contracts (so a gpu failure is a kernel bug)."""
    for f in (gc.check_gemm, gc.check_conv, gc.check_norms, gc.check_attention, gc.check_elementwise):
        _ok(f())


def test_conditioning_cache_is_reused_and_invalidated(cpu_ops):
    native, _, ocfg = gc.build_pair("mini", 1234)
    inp = gc.config1_inputs(ocfg, 1, 4, 8)
    kw = dict(fps=inp["fps"], image_latents=inp["image_latents"].half(), image_embeddings=inp["image_embeddings"].half(),
              encoder_hidden_states=inp["encoder_hidden_states"].half())
    v1 = native(inp["sample"].half(), 981, **kw)[0]
    ctx = native._ctx
    v2 = native(inp["sample"].half(), 961, **kw)[0]
    assert native._ctx is ctx, "step-invariant conditioning must be computed once per clip"
    assert not torch.allclose(v1, v2)
    kw["encoder_hidden_states"].mul_(2.0)  # in-place change bumps the version counter -> cache miss
    native(inp["sample"].half(), 981, **kw)
    assert native._ctx is not ctx


def test_product_schedulers_vs_oracle_and_golden(cpu_ops):
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    from oracle import schedulers_oracle as so
    g = torch.load(os.path.join(gc.ROOT, "tests", "golden", "inverse_scheduler.pt"))
    inv, fwd = DDIMInverseScheduler.from_pretrained("ali-vilab/i2vgen-xl", subfolder="scheduler"), DDIMScheduler()
    np.testing.assert_allclose(inv.alphas_cumprod.numpy(), g["alphas_cumprod"].numpy(), rtol=2e-6, atol=1e-9)
    inv.set_timesteps(500)
    assert inv.timesteps.tolist() == g["timesteps_500"].tolist() == list(range(1, 1000, 2))
    fwd.set_timesteps(50)
    assert fwd.timesteps.tolist() == list(range(981, 0, -20))  # demo.ipynb:1201-1204
    inv.set_timesteps(50)
    ac = so.alphas_cumprod()
    x, v = g["x"], g["v"]
    for t in (1, 21, 501, 981):
        got = inv.step(v.half(), t, x.half()).prev_sample.float().numpy()
        np.testing.assert_allclose(got, so.inverse_step(v.numpy(), t, x.numpy(), 50, ac), rtol=0, atol=4e-3)
        np.testing.assert_allclose(got, g[f"inv_step_n50_t{t}"].numpy(), rtol=0, atol=4e-3)
        got = fwd.step(v.half(), t, x.half()).prev_sample.float().numpy()
        np.testing.assert_allclose(got, so.ddim_step(v.numpy(), t, x.numpy(), 50, ac), rtol=0, atol=4e-3)
    assert fwd.init_noise_sigma == 1.0 and fwd.order == 1
    assert fwd.scale_model_input(x, 5) is x
    with pytest.raises(ValueError):
        DDIMScheduler().step(v, 1, x)  # set_timesteps not called


def test_init_pnp_schedule_semantics():
    """int() truncation and prefixes of the FULL list (run_group_pnp_edit.py:36-45): int(50*0.29) == 14."""
    from types import SimpleNamespace
    from anyv2v_amd import pnp_utils
    from anyv2v_amd.run_group_pnp_edit import init_pnp, output_suffix
    from anyv2v_amd.schedulers import DDIMScheduler
    from anyv2v_amd.unet import I2VGenXLUNet, I2VGenXLUNetConfig, pnp_on
    with torch.device("meta"):
        unet = I2VGenXLUNet(I2VGenXLUNetConfig.mini())
    pipe = SimpleNamespace(unet=unet)
    s = DDIMScheduler()
    s.set_timesteps(50)
    cfg = SimpleNamespace(n_steps=50, pnp_f_t=0.29, pnp_spatial_attn_t=0.2, pnp_temp_attn_t=1.0, cfg=9.0)
    init_pnp(pipe, s, cfg)
    r = unet.up_blocks[1].resnets[1]
    assert len(r.injection_schedule) == 14 and max(r.injection_schedule) == 981 and min(r.injection_schedule) == 981 - 13 * 20
    sp = unet.up_blocks[2].attentions[0].transformer_blocks[0].attn1.processor
    tp = unet.up_blocks[3].temp_attentions[2].transformer_blocks[0].attn1.processor
    assert len(sp.injection_schedule) == 10 and len(tp.injection_schedule) == 50
    # sites: not block 0 of up_blocks[1] (pnp_utils.py:235)
    assert unet.up_blocks[1].attentions[0].transformer_blocks[0].attn1.processor.injection_schedule is None
    pnp_utils.register_time(pipe, 801)
    # (conv, 8 spatial sites, 8 temporal sites): every site contributes to the HIP-graph key
    assert pnp_utils.injection_state(pipe) == (True,) + (True,) * 8 + (True,) * 8
    pnp_utils.register_time(pipe, 781)
    assert pnp_utils.injection_state(pipe) == (True,) + (False,) * 8 + (True,) * 8
    pnp_utils.register_time(pipe, 701)
    assert pnp_utils.injection_state(pipe) == (False,) + (False,) * 8 + (True,) * 8
    # one site with its own schedule ("Disable PNP" per module, pnp_utils.py:229-232) changes the key
    one = unet.up_blocks[3].attentions[1].transformer_blocks[0].attn1.processor
    saved, one.injection_schedule = one.injection_schedule, frozenset()
    pnp_utils.register_time(pipe, 801)
    st = pnp_utils.injection_state(pipe)
    assert st != (True,) * 17 and sum(st) == 16
    one.injection_schedule = saved
    assert pnp_on(1000, frozenset()) and not pnp_on(999, frozenset())  # magic t == 1000 (pnp_utils.py:109)
    pnp_utils.clear_time(pipe)
    assert pnp_utils.injection_state(pipe) == (False,) * 17
    assert output_suffix(cfg, 0) == "ddim_init_latents_t_idx_0_nsteps_50_cfg_9.0_pnpf0.29_pnps0.2_pnpt1.0"


def test_latent_trajectory_store_roundtrip(tmp_path):
    from anyv2v_amd.utils import LatentTrajectory, load_ddim_latents_at_T, load_ddim_latents_at_t
    tr = LatentTrajectory()
    for t in (1, 21, 981):
        tr[t] = torch.full((1, 4, 2, 3, 3), float(t), dtype=torch.float16)
    d = tmp_path / "ddim_latents"
    tr.save(str(d))
    tr.wait()
    assert sorted(os.listdir(d)) == ["ddim_latents_1.pt", "ddim_latents_21.pt", "ddim_latents_981.pt"]  # utils.py:26
    x = load_ddim_latents_at_t(21, str(d))
    assert x.dtype == torch.float16 and tuple(x.shape) == (1, 4, 2, 3, 3) and float(x[0, 0, 0, 0, 0]) == 21
    assert float(load_ddim_latents_at_T(str(d)).max()) == 981
    with pytest.raises(AssertionError, match="Missing latents"):
        load_ddim_latents_at_t(41, str(d))
    tr2 = LatentTrajectory.load(str(d))
    assert sorted(tr2.keys()) == [1, 21, 981]


def test_background_trajectory_writer_never_races_its_readers(tmp_path, monkeypatch):
    """invert(output_dir=d) then sample_with_pnp(ddim_inv_latents_path=d) in one process.  With the background
    writer every reader joins it first; files appear atomically; a fresh directory appears only when complete."""
    import time
    from anyv2v_amd import utils
    from anyv2v_amd.utils import LatentTrajectory, inversion_is_complete, load_ddim_latents_at_T, load_ddim_latents_at_t
    real_save = torch.save

    def slow_save(obj, path, *a, **k):
        time.sleep(0.05)
        return real_save(obj, path, *a, **k)

    monkeypatch.setattr(utils.torch, "save", slow_save)
    tr = LatentTrajectory()
    for t in range(1, 400, 20):
        tr[t] = torch.full((1, 4, 2, 3, 3), float(t), dtype=torch.float16)
    out = tmp_path / "exp"
    d = out / "ddim_latents"
    tr.save(str(d), background=True)
    assert not d.exists() and not inversion_is_complete(str(out), str(d))   # staged under *.partial-<pid> until complete
    assert float(load_ddim_latents_at_t(381, str(d))[0, 0, 0, 0, 0]) == 381      # joins the writer instead of asserting
    assert d.exists() and inversion_is_complete(str(out), str(d)) and not utils._PENDING
    assert len(os.listdir(d)) == 20 and all(n.startswith("ddim_latents_") for n in os.listdir(d))
    tr.save(str(d), background=True)                                                # directory exists: per-file atomic replace
    assert float(load_ddim_latents_at_T(str(d)).max()) == 381
    assert len(LatentTrajectory.load(str(d))) == 20
    # a crashed writer's leftovers make the entry "not complete" and are cleaned by the next save
    os.makedirs(str(d) + ".partial-99999")
    assert not inversion_is_complete(str(out), str(d))
    tr.save(str(d))                                                                 # default: synchronous
    assert inversion_is_complete(str(out), str(d))


def test_one_clip_per_call_and_crop_rounding(cpu_ops):
    from PIL import Image
    from anyv2v_amd.encoders import _center_crop_wide
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from anyv2v_amd.schedulers import DDIMInverseScheduler
    native, _, ocfg = gc.build_pair("mini", 1234)
    pipe = I2VGenXLPipeline(unet=native, scheduler=DDIMInverseScheduler())
    inp = gc.config1_inputs(ocfg, 2, 4, 8)
    kw = dict(image_embeddings=inp["image_embeddings"][:1].half(), image_latents=inp["image_latents"][:1].half(), height=64,
              width=64, num_frames=4, num_inference_steps=2, guidance_scale=1.0, target_fps=8, latents=inp["sample"][:1].half())
    with pytest.raises(ValueError, match="one clip per call"):
        pipe.invert(prompt_embeds=inp["encoder_hidden_states"].half(), **kw)           # a batch of two prompts
    with pytest.raises(ValueError, match="num_videos_per_prompt"):
        pipe.invert(prompt_embeds=inp["encoder_hidden_states"][:1].half(), num_videos_per_prompt=2, **kw)
    # pipeline_i2vgen_xl.py:1496,1505: round(width // scale) -- 1003x600 -> 512x512: scale 1.171875, 1003 // scale = 855.0
    # (1003 / scale = 855.89 would round to 856 and shift the crop window)
    img = Image.fromarray((np.arange(600 * 1003 * 3) % 251).astype(np.uint8).reshape(600, 1003, 3))
    scale = min(1003 / 512, 600 / 512)
    ref = img.resize((round(1003 // scale), round(600 // scale)), resample=Image.BOX)
    x1, y1 = (ref.width - 512) // 2, (ref.height - 512) // 2
    assert ref.width == 855
    assert np.array_equal(np.asarray(_center_crop_wide(img, (512, 512))), np.asarray(ref.crop((x1, y1, x1 + 512, y1 + 512))))


def test_pipeline_input_checks():
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    p = I2VGenXLPipeline()
    with pytest.raises(ValueError, match="divisible by 8"):
        p.check_inputs("a", None, 100, 512)
    with pytest.raises(ValueError, match="Cannot forward both"):
        p.check_inputs("a", None, 512, 512, prompt_embeds=torch.zeros(1))
    with pytest.raises(ValueError, match="Provide either"):
        p.check_inputs(None, None, 512, 512)
    with pytest.raises(FileNotFoundError):
        I2VGenXLPipeline.from_pretrained("ali-vilab/i2vgen-xl")  # no weights, no seed -> loud
    # options the reference hands to the forward DDIM step / the attention processors: refused, not dropped
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    p.scheduler = DDIMScheduler()
    p._forward_sampler_options(0.0, None)
    p._forward_sampler_options(None, {})
    with pytest.raises(ValueError, match="eta"):
        p._forward_sampler_options(0.5, None)
    with pytest.raises(ValueError, match="cross_attention_kwargs"):
        p._forward_sampler_options(0.0, {"scale": 0.5})
    p.scheduler = DDIMInverseScheduler()
    p._forward_sampler_options(0.5, None)          # (the reference's inverse scheduler takes no eta: dropped there too)


def test_output_types_follow_the_references_tensor2vid():
    """``output_type`` "pil" / "np" / "pt" / "latent" (``pipeline_i2vgen_xl.py:79-97,876-880``): the reference's own ``tensor2vid``
    (its file imported verbatim) on the same decoded video vs ``I2VGenXLPipeline._finish``; anything else is refused as there."""
    from oracle import ref_stubs
    if not ref_stubs.reference_available():
        pytest.skip("needs /root/reference")
    from anyv2v_amd.encoders import SyntheticVAE
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from oracle import ref_pipeline
    pm = ref_pipeline.load_reference_pipeline_module()[0]
    pipe = I2VGenXLPipeline(vae=SyntheticVAE())
    lat = torch.randn(1, 4, 3, 4, 6, generator=torch.Generator().manual_seed(0)).half()
    video = pipe.decode_latents(lat, decode_chunk_size=1)
    proc = pm.VaeImageProcessor(vae_scale_factor=8, do_resize=False)
    want_np = pm.tensor2vid(video.float(), proc, "np")
    got_np = pipe._finish(lat, "np", 1, True).frames
    assert isinstance(got_np, np.ndarray) and got_np.dtype == np.float32 and got_np.shape == want_np.shape == (1, 3, 32, 48, 3)
    assert np.allclose(got_np, want_np, atol=1e-6)
    want_pt = pm.tensor2vid(video.float(), proc, "pt")
    got_pt = pipe._finish(lat, "pt", 1, False)[0]
    assert torch.is_tensor(got_pt) and got_pt.shape == want_pt.shape and torch.allclose(got_pt, want_pt, atol=1e-6)
    want_pil = pm.tensor2vid(video.float(), proc, "pil")
    got_pil = pipe._finish(lat, "pil", 1, True).frames
    assert len(got_pil) == 1 and len(got_pil[0]) == 3
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(got_pil[0], want_pil[0]))
    assert pipe._finish(lat, "latent", 1, True).frames is lat
    with pytest.raises(ValueError, match="does not exist"):
        pipe._finish(lat, "mp4", 1, True)
    with pytest.raises(ValueError, match="does not exist"):
        pm.tensor2vid(video.float(), proc, "mp4")
    # the module-level function, with and without a processor
    from anyv2v_amd.pipeline import tensor2vid
    assert np.allclose(tensor2vid(video, None, "np"), want_np, atol=1e-6)
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(tensor2vid(video, None, "pil")[0], want_pil[0]))
    # the rest of the reference pipeline's public surface
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    pipe.scheduler = DDIMScheduler()
    assert pipe.prepare_extra_step_kwargs("g", 0.3) == {"eta": 0.3, "generator": "g"}
    pipe.scheduler = DDIMInverseScheduler()
    assert pipe.prepare_extra_step_kwargs("g", 0.3) == {}          # (``consisti2v/ddim_inverse_scheduler.py:291-297`` takes neither)
    for name in ("enable_vae_slicing", "disable_vae_slicing", "enable_vae_tiling", "disable_vae_tiling", "disable_freeu"):
        getattr(pipe, name)()
    with pytest.raises(NotImplementedError):
        pipe.enable_freeu(0.9, 0.2, 1.2, 1.4)
    pipe._guidance_scale = 9.0
    x = torch.rand(1, 3, 32, 48, generator=torch.Generator().manual_seed(1)) * 2 - 1
    il = pipe.prepare_image_latents(x, "cpu", 4, 1)
    assert tuple(il.shape) == (2, 4, 4, 4, 6) and torch.equal(il[0], il[1])
    assert torch.allclose(il[0, :, 1].float(), torch.full((4, 4, 6), 1 / 3), atol=1e-3) and torch.allclose(il[0, :, 3].float(), torch.ones(4, 4, 6))
    pipe._guidance_scale = 1.0
    assert tuple(pipe.prepare_image_latents(x, "cpu", 4, 1).shape) == (1, 4, 4, 4, 6)


def test_load_image_applies_exif_orientation_and_takes_pil_images(tmp_path):
    """[3P] ``diffusers.utils.load_image``: path or PIL image, EXIF orientation applied, RGB."""
    from PIL import Image
    from anyv2v_amd.utils import load_image
    img = Image.fromarray((np.arange(6 * 4 * 3) % 255).astype(np.uint8).reshape(6, 4, 3))      # 4 wide, 6 high
    exif = Image.Exif()
    exif[0x0112] = 6                                                                            # "rotate 90 CW to display"
    p = str(tmp_path / "rot.jpg")
    img.save(p, exif=exif, quality=100)
    got = load_image(p)
    assert got.mode == "RGB" and got.size == (6, 4)                                             # displayed orientation: 6 wide, 4 high
    assert load_image(img.convert("L")).mode == "RGB" and load_image(img).size == (4, 6)
    with pytest.raises(ValueError):
        load_image(str(tmp_path / "missing.png"))
    with pytest.raises(ValueError):
        load_image(3)


def test_native_vae_host_logic_and_state_dict(cpu_ops):
    """AutoencoderKL wiring over the token layout (CPU emulation of the ops): diffusers state-dict keys / shapes of the
    full SD-VAE, mini-config encode / decode vs the oracle, and the pipeline-facing adapter."""
    import numpy as np
    from PIL import Image

    from anyv2v_amd.encoders import NativeVAE
    from anyv2v_amd.vae import AutoencoderKL, VAEConfig
    from oracle import vae_oracle as vo
    full, full_o = AutoencoderKL(), vo.AutoencoderKLOracle()
    assert list(full.state_dict()) == list(full_o.state_dict()) and len(full.state_dict()) == 248
    assert all(full.state_dict()[k].shape == v.shape for k, v in full_o.state_dict().items())
    assert sum(p.numel() for p in full_o.parameters()) == 83_653_863  # the SD-VAE's parameter count
    _ok(gc.check_vae(full=False))
    v = NativeVAE(random_init_seed=1, cfg=VAEConfig.mini())
    imgs = [Image.fromarray((np.random.RandomState(i).rand(40, 60, 3) * 255).astype("uint8")) for i in range(3)]
    lat = v.encode_video(imgs, torch.device("cpu"), 32, 48)
    assert lat.shape == (1, 4, 3, 16, 24) and lat.dtype == torch.float16
    vid = v.decode_video(lat, decode_chunk_size=2)
    assert vid.shape == (1, 3, 3, 32, 48) and float(vid.abs().max()) <= 1.0
    assert len(v.to_pil(vid)) == 3


def test_hf_clip_adapters_follow_the_reference_encode_paths():
    """``HFTextEncoder`` / ``HFImageEncoder`` on tiny randomly initialised CLIP towers: clip_skip semantics
    (hidden state -(clip_skip+1) + final LayerNorm, pipeline_i2vgen_xl.py:312-324) and the image-embedding shape."""
    transformers = pytest.importorskip("transformers")
    from PIL import Image

    from hf_clip_reference import HFImageEncoder, HFTextEncoder
    tm = transformers.CLIPTextModel(transformers.CLIPTextConfig(vocab_size=100, hidden_size=32, intermediate_size=64,
                                                                num_hidden_layers=3, num_attention_heads=4,
                                                                max_position_embeddings=16, bos_token_id=1, eos_token_id=2))

    class Tok:
        model_max_length = 16

        def __call__(self, prompts, padding, max_length, truncation, return_tensors):
            ids = torch.zeros(len(prompts), max_length, dtype=torch.long)
            for i, p in enumerate(prompts):
                t = [(ord(c) % 90) + 3 for c in p][: max_length - 1]
                ids[i, : len(t)] = torch.tensor(t, dtype=torch.long)
                ids[i, len(t)] = 2
            return type("O", (), {"input_ids": ids})

    enc = HFTextEncoder(tm, Tok())
    e = enc.encode(["a robot", ""], torch.device("cpu"), clip_skip=1)
    assert e.shape == (2, 16, 32) and e.dtype == torch.float16
    ids = Tok()(["a robot", ""], "max_length", 16, True, "pt").input_ids
    hs = tm(ids, output_hidden_states=True).hidden_states
    ln = getattr(tm, "text_model", tm).final_layer_norm
    assert torch.allclose(e.float(), ln(hs[-2]).detach(), atol=2e-3)
    assert torch.allclose(enc.encode("a robot", torch.device("cpu"), None).float(), tm(ids[:1])[0].detach(), atol=2e-3)
    vm = transformers.CLIPVisionModelWithProjection(transformers.CLIPVisionConfig(
        hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, image_size=224, patch_size=32,
        projection_dim=24))
    img = Image.fromarray((np.random.RandomState(0).rand(300, 500, 3) * 255).astype("uint8"))
    assert HFImageEncoder(vm).encode(img, 512, torch.device("cpu")).shape == (1, 1, 24)


def _tiny_clip_pair(transformers, head_dim_text=64, head_dim_vis=80):
    """Randomly initialised tiny CLIP towers with the head dims of the checkpoint's (text 64, vision 80) + their state dicts."""
    torch.manual_seed(0)
    tcfg = transformers.CLIPTextConfig(vocab_size=100, hidden_size=2 * head_dim_text, intermediate_size=256, num_hidden_layers=3,
                                       num_attention_heads=2, max_position_embeddings=16, bos_token_id=1, eos_token_id=2,
                                       hidden_act="gelu")
    vcfg = transformers.CLIPVisionConfig(hidden_size=2 * head_dim_vis, intermediate_size=320, num_hidden_layers=2,
                                         num_attention_heads=2, image_size=224, patch_size=32, projection_dim=24, hidden_act="gelu")
    tm, vm = transformers.CLIPTextModel(tcfg).eval(), transformers.CLIPVisionModelWithProjection(vcfg).eval()
    with torch.no_grad():  # the default init leaves most activations ~0: make every layer matter
        for m in (tm, vm):
            for n_, p_ in m.named_parameters():
                if p_.dim() >= 2:
                    p_.normal_(0, 0.08)
                elif "bias" in n_:
                    p_.normal_(0, 0.05)
    return tm, vm, tcfg, vcfg


def test_native_clip_towers_vs_transformers(cpu_ops):
    """``anyv2v_amd.clip`` (the towers behind encode_prompt / _encode_image, pipeline_i2vgen_xl.py:224-441) on the
    emulated ops vs ``transformers``' own CLIPTextModel / CLIPVisionModelWithProjection in fp32 on the same weights: every hidden
    state (causal mask), clip_skip in {None, 1, 2} through the final LayerNorm, and the projected image embedding (head_dim 80,
    class token, pre / post LayerNorm).  The same comparison runs on the real kernels in the -m gpu suite."""
    transformers = pytest.importorskip("transformers")
    from anyv2v_amd.clip import CLIPTextTower, CLIPTowerConfig, CLIPVisionTower
    tm, vm, tcfg, vcfg = _tiny_clip_pair(transformers)
    ids = torch.randint(3, 100, (3, 16))
    ids[:, -3:] = 2
    text = CLIPTextTower(CLIPTowerConfig.from_hf(tcfg.to_dict()), tm.state_dict()).to("cpu")
    with torch.no_grad():
        want = tm(ids, output_hidden_states=True)
    got = text.hidden_states(ids)
    assert len(got) == len(want.hidden_states) == 4
    for g_, w_ in zip(got, want.hidden_states):
        assert (g_.float() - w_).abs().max() <= 6e-3 * max(1.0, float(w_.abs().max()))
    ln = getattr(tm, "text_model", tm).final_layer_norm
    for skip in (None, 1, 2):
        ref = (want.last_hidden_state if skip is None else ln(want.hidden_states[-(skip + 1)])).detach()
        assert (text.encode_ids(ids, skip).float() - ref).abs().max() <= 8e-3 * float(ref.abs().max())
    vis = CLIPVisionTower(CLIPTowerConfig.from_hf(vcfg.to_dict()), vm.state_dict()).to("cpu")
    px = torch.randn(2, 3, 224, 224)
    with torch.no_grad():
        ref = vm(pixel_values=px).image_embeds
    got = vis.image_embeds(px)
    assert got.shape == (2, 24) and (got.float() - ref).abs().max() <= 8e-3 * float(ref.abs().max())


def test_native_clip_encoders_follow_the_reference_encode_paths(cpu_ops, tmp_path):
    """``attach_native_clip_encoders`` on a checkpoint-shaped folder (text_encoder / tokenizer / image_encoder with config.json +
    model.safetensors): the encoders it builds agree with the transformers-module adapters on the reference's encode paths."""
    transformers = pytest.importorskip("transformers")
    from PIL import Image
    from safetensors.torch import save_file

    from anyv2v_amd.encoders import NativeImageEncoder, NativeTextEncoder, _load_tower_files
    from hf_clip_reference import HFImageEncoder, HFTextEncoder
    from anyv2v_amd.clip import CLIPTextTower, CLIPTowerConfig, CLIPVisionTower
    tm, vm, tcfg, vcfg = _tiny_clip_pair(transformers)
    for name, m, c in (("text_encoder", tm, tcfg), ("image_encoder", vm, vcfg)):
        d = tmp_path / name
        d.mkdir()
        (d / "config.json").write_text(c.to_json_string())
        save_file({k: v.contiguous() for k, v in m.state_dict().items()}, str(d / "model.safetensors"))

    class Tok:
        model_max_length = 16

        def __call__(self, prompts, padding, max_length, truncation, return_tensors):
            ids = torch.zeros(len(prompts), max_length, dtype=torch.long)
            for i, p in enumerate(prompts):
                t = [(ord(c) % 90) + 3 for c in p][: max_length - 1]
                ids[i, : len(t)] = torch.tensor(t, dtype=torch.long)
                ids[i, len(t)] = 2
            return type("O", (), {"input_ids": ids})

    tc, tsd = _load_tower_files(str(tmp_path / "text_encoder"))
    vc, vsd = _load_tower_files(str(tmp_path / "image_encoder"))
    nat_t = NativeTextEncoder(CLIPTextTower(CLIPTowerConfig.from_hf(tc, "text"), tsd), Tok())
    nat_v = NativeImageEncoder(CLIPVisionTower(CLIPTowerConfig.from_hf(vc, "vision"), vsd))
    dev = torch.device("cpu")
    for skip in (None, 1):
        a = nat_t.encode(["a robot", ""], dev, clip_skip=skip)
        b = HFTextEncoder(tm, Tok()).encode(["a robot", ""], dev, clip_skip=skip)
        assert a.shape == b.shape == (2, 16, 128) and (a.float() - b.float()).abs().max() <= 8e-3 * float(b.float().abs().max())
    img = Image.fromarray((np.random.RandomState(0).rand(300, 500, 3) * 255).astype("uint8"))
    a, b = nat_v.encode(img, 512, dev), HFImageEncoder(vm).encode(img, 512, dev)
    assert a.shape == b.shape == (1, 1, 24) and (a.float() - b.float()).abs().max() <= 8e-3 * float(b.float().abs().max())


def test_pnp_without_cfg_is_refused(cpu_ops):
    """Reference quirk B.1: with guidance_scale <= 1 the hooks would slice a 2-way batch in thirds; we raise instead."""
    from anyv2v_amd import pnp_utils
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from anyv2v_amd.schedulers import DDIMScheduler
    from anyv2v_amd.utils import LatentTrajectory
    native, _, ocfg = gc.build_pair("mini", 7)
    pipe = I2VGenXLPipeline(unet=native, scheduler=DDIMScheduler())
    pipe._device = torch.device("cpu")
    inp = gc.config1_inputs(ocfg, 3, 4, 8)
    sched = DDIMScheduler()
    sched.set_timesteps(4)
    pnp_utils.register_conv_injection(pipe, sched.timesteps)
    pnp_utils.register_spatial_attention_pnp(pipe, sched.timesteps)
    pnp_utils.register_temp_attention_pnp(pipe, sched.timesteps)
    traj = LatentTrajectory()
    for t in sched.timesteps.tolist():
        traj[int(t)] = inp["sample"][:1].half()
    h = lambda x: x.half()
    with pytest.raises(ValueError, match="classifier-free guidance"):
        pipe.sample_with_pnp(prompt_embeds=h(inp["encoder_hidden_states"][2:3]), image_embeddings=h(inp["image_embeddings"][2:3]),
                             image_latents=h(inp["image_latents"][2:3]), height=64, width=64, num_frames=4,
                             num_inference_steps=4, guidance_scale=1.0, target_fps=8, latents=h(inp["sample"][:1]),
                             output_type="latent", ddim_init_latents_t_idx=0, ddim_inv_latents_path=traj,
                             ddim_inv_prompt_embeds=h(inp["encoder_hidden_states"][:1]),
                             ddim_inv_image_embeddings=h(inp["image_embeddings"][:1]),
                             ddim_inv_image_latents=h(inp["image_latents"][:1]))
    pnp_utils.clear_time(pipe)


# ---- the REFERENCE's own pipeline class, imported verbatim, runs stage 1 and stage 2 on the CPU ------------------------------
def test_reference_pipeline_code_vs_loop_oracle_vs_native_pipeline(cpu_ops, tmp_path):
    """A1 / A2 / A3 / A12 pinned to the reference's code: ``/root/reference/i2vgen-xl/pipelines/pipeline_i2vgen_xl.py`` (verbatim,
    via ``oracle.ref_pipeline``; its ``invert`` :1197-1451, ``sample_with_pnp`` :892-1193 and ``__call__`` :652-888 loops, its
    ``encode_prompt`` / ``_encode_image`` / ``prepare_image_latents`` / ``encode_vae_video`` glue, its ``pnp_utils.py`` hooks and
    ``utils.load_ddim_latents_at_t``) runs 4-step inversion -> CFG reconstruction -> PnP edit of a 4-frame 64x64 clip around the
    oracle UNet, the reference's vendored inverse scheduler and toy VAE / CLIP components.  Compared with (a) ``oracle.pnp_oracle``'s
    loops on the same conditioning tensors -- the loop oracle the GPU tests use -- and (b) the native pipeline on the same weights
    (op emulation, fp16): trajectory files, edited latents, reconstructed latents.  (c) The committed fixture
    ``tests/golden/ref_pipeline_mini.pt`` is what this produces (the -m gpu suite compares the HIP path with it)."""
    from oracle import ref_stubs
    if not ref_stubs.reference_available():
        pytest.skip("needs /root/reference")
    import copy
    import sys
    from oracle import pnp_oracle, ref_pipeline
    sys.path.insert(0, os.path.join(gc.ROOT, "tests", "golden"))
    import make_golden
    spec = make_golden.REF_PIPELINE_JOBS["mini"]
    _, oracle, ocfg = gc.build_pair("mini", spec["seed"])
    oracle_plain = copy.deepcopy(oracle)  # for the loop oracle: the reference hooks replace forwards / processors in place
    frames, edited = make_golden.ref_pipeline_frames(spec)
    n_steps, size, ratios = spec["n_steps"], spec["size"], spec["ratios"]
    job = ref_pipeline.run_reference_job(oracle, ocfg.cross_attention_dim, frames, edited, size, n_steps, ratios, tmp_path)
    files, inv_ts, T = job["files"], job["inv_ts"], job["T"]
    assert torch.equal(job["inverted"][0, 0], files[inv_ts[-1]][0])  # reversed stack: index 0 = the noisiest latent (:1436)
    with torch.no_grad():
        cond_src = dict(fps=torch.tensor([8]), image_latents=job["src_il"], image_embeddings=job["src_ie"],
                        encoder_hidden_states=job["src_pe"])
        traj_o = pnp_oracle.invert_loop(oracle_plain, job["lat0"].clone(), cond_src, n_steps)
        for t in inv_ts:
            assert (traj_o[t] - files[t]).abs().max() <= 2e-4 * files[t].abs().max(), f"loop oracle vs reference invert at t={t}"
        cond2 = dict(fps=torch.tensor([8, 8]), image_latents=torch.cat([job["src_il"], job["src_il"]]),
                     image_embeddings=torch.cat([torch.zeros_like(job["src_ie"]), job["src_ie"]]),
                     encoder_hidden_states=torch.cat([job["rec_npe"], job["rec_pe"]]))
        rec_o = pnp_oracle.sample_loop(oracle_plain, files[T].clone(), cond2, n_steps, 9.0, t_idx=0)
        assert (rec_o - job["rec_ref"]).abs().max() <= 5e-4 * job["rec_ref"].abs().max(), "loop oracle vs reference __call__"
        cond_all = dict(fps=torch.tensor([8, 8, 8]), image_latents=torch.cat([job["src_il"], job["il2"]]),
                        image_embeddings=torch.cat([job["src_ie"], job["ie2"]]),
                        encoder_hidden_states=torch.cat([job["src_pe"], job["npe"], job["pe"]]))
        plain_o = pnp_oracle.pnp_loop(oracle_plain, files[T].clone(), files, cond_all, n_steps, 9.0, t_idx=0)  # no hooks yet
        pnp_oracle.init_pnp(oracle_plain, n_steps, *ratios)
        edit_o = pnp_oracle.pnp_loop(oracle_plain, files[T].clone(), files, cond_all, n_steps, 9.0, t_idx=0)
        pnp_oracle.clear_hooks(oracle_plain)
        edit_ref = job["edit_ref"]
        assert (edit_o - edit_ref).abs().max() <= 5e-4 * edit_ref.abs().max(), "loop oracle vs reference sample_with_pnp"
        # not vacuous: the edit moved away from its start, and the hooks mattered
        assert (edit_ref - files[T]).abs().max() > 0.5 * files[T].abs().max()
        assert (plain_o - edit_ref).abs().max() > 20 * (edit_o - edit_ref).abs().max()
    # (c) the committed fixture is this job
    fx = make_golden.pack_ref_pipeline_job(spec, job)
    gold = torch.load(os.path.join(gc.ROOT, "tests", "golden", "ref_pipeline_mini.pt"))
    for k in ("trajectory", "edit_ref", "rec_ref", "src_pe", "il_edit"):
        assert (fx[k].float() - gold[k].float()).abs().max() <= 2e-3 * gold[k].float().abs().max().clamp_min(1e-6), k
    # (b) the native pipeline (fp16, emulated ops) against the reference-generated fixture -- the same check the GPU suite runs
    _ok(gc.check_pipeline_vs_reference_fixture("mini"))


def test_native_pipeline_from_raw_inputs_vs_reference_pipeline(cpu_ops, tmp_path):
    """A13 (once-per-clip pre / post) pinned to the reference's glue: BOTH pipelines start from the same PIL frames and prompt
    strings with the same toy VAE / CLIP weights -- the reference's ``encode_vae_video``, ``encode_prompt`` (clip_skip = 1,
    negative prompt), ``_encode_image`` (centre crop, 224 bilinear, CLIP normalisation, zero negative embedding),
    ``prepare_image_latents`` (scaling factor, frame-position planes) vs ``anyv2v_amd.pipeline`` + ``anyv2v_amd.encoders``'
    interfaces around the same components -- and must agree on every conditioning tensor and on the final latents."""
    from oracle import ref_stubs
    if not ref_stubs.reference_available():
        pytest.skip("needs /root/reference")
    import sys
    from PIL import Image

    from anyv2v_amd import pnp_utils
    from anyv2v_amd.encoders import _center_crop_wide, _pil_to_tensor
    from hf_clip_reference import HFImageEncoder, HFTextEncoder
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from anyv2v_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    from oracle import ref_pipeline
    sys.path.insert(0, os.path.join(gc.ROOT, "tests", "golden"))
    import make_golden
    spec = dict(make_golden.REF_PIPELINE_JOBS["mini"], size=128)  # 128 x 128: not the size the fixtures use
    frames, edited = make_golden.ref_pipeline_frames(spec)
    frames = [f.resize((200, 128)) for f in frames]                # wide source frames: the centre crop matters
    edited = edited.resize((160, 128))
    native, oracle, ocfg = gc.build_pair("mini", spec["seed"])
    n_steps, size, ratios = spec["n_steps"], spec["size"], spec["ratios"]
    job = ref_pipeline.run_reference_job(oracle, ocfg.cross_attention_dim, frames, edited, size, n_steps, ratios, tmp_path,
                                         with_reconstruction=False)

    class VaeAdapter:  # the toy VAE behind anyv2v_amd.encoders' VAE interface (what NativeVAE does around the real AutoencoderKL)
        def __init__(self):
            self.toy = ref_pipeline.ToyVAE()
            self.config = self.toy.config

        def to(self, device):
            return self

        def _enc(self, x):
            return (self.toy.encode(x).latent_dist.sample() * self.config.scaling_factor).detach()

        def encode_image(self, image, device, height, width):
            return self._enc(_pil_to_tensor(_center_crop_wide(image, (width, height)))).half()

        def encode_video(self, video, device, height, width):
            x = torch.cat([_pil_to_tensor(_center_crop_wide(f, (width, height))) for f in video])
            return self._enc(x).permute(1, 0, 2, 3)[None].half()

    dim, dev = ocfg.cross_attention_dim, torch.device("cpu")
    tok = ref_pipeline.ToyTokenizer()
    pipe = I2VGenXLPipeline(unet=native, scheduler=DDIMInverseScheduler(), vae=VaeAdapter(),
                            text_encoder=HFTextEncoder(ref_pipeline.ToyTextEncoder(dim), tok), tokenizer=tok,
                            image_encoder=HFImageEncoder(ref_pipeline.ToyImageEncoder(dim)), feature_extractor=object())
    pipe._device = dev
    close = lambda a, b, tol: float((a.float() - b.float()).abs().max()) <= tol * float(b.float().abs().max())
    lat0 = pipe.encode_vae_video(frames, dev, height=size, width=size)
    assert close(lat0, job["lat0"], 2e-3), "encode_vae_video"
    assert close(pipe.text_encoder.encode("", dev, 1), job["src_pe"], 2e-3) and close(pipe.text_encoder.encode(job["neg"], dev, 1), job["npe"], 2e-3)
    assert close(pipe.image_encoder.encode(frames[0], size, dev), job["src_ie"], 2e-3), "_encode_image (crop + 224 bilinear + CLIP norm)"
    first = pipe.vae.encode_image(edited, dev, size, size)
    assert close(pipe.prepare_image_latents_from_first_frame_latent(first, len(frames)), job["il2"][1:], 2e-3), "prepare_image_latents"
    traj = pipe.invert(prompt="", image=frames[0], height=size, width=size, num_frames=len(frames), num_inference_steps=n_steps,
                       guidance_scale=1.0, negative_prompt=job["neg"], target_fps=8, latents=lat0, return_trajectory=True)
    for t in job["inv_ts"]:
        assert close(traj[t], job["files"][t], 1e-2), f"invert from raw inputs at t={t}"
    sched = DDIMScheduler()
    sched.set_timesteps(n_steps)
    pipe.register_modules(scheduler=sched)
    k = lambda r: sched.timesteps[: int(n_steps * r)]
    pnp_utils.register_conv_injection(pipe, k(ratios[0]))
    pnp_utils.register_spatial_attention_pnp(pipe, k(ratios[1]))
    pnp_utils.register_temp_attention_pnp(pipe, k(ratios[2]))
    T = job["T"]
    ed = pipe.sample_with_pnp(prompt="a robot", image=edited, height=size, width=size, num_frames=len(frames),
                              num_inference_steps=n_steps, guidance_scale=9.0, negative_prompt=job["neg"], target_fps=8,
                              latents=job["files"][T].half(), output_type="latent", ddim_init_latents_t_idx=0,
                              ddim_inv_latents_path=job["out_dir"], ddim_inv_prompt="", ddim_inv_1st_frame=frames[0]).frames
    pnp_utils.clear_time(pipe)
    assert close(ed, job["edit_ref"], 3e-2), "sample_with_pnp from raw inputs (reads the reference's ddim_latents_{t}.pt files)"


def test_source_feature_cache_byte_budget_and_trajectory_serials():
    """the multi-edit cache has a byte budget (a step that does not fit is not recorded -> it stays a three-branch step)
    and is keyed on a process-unique trajectory serial, not on ``id()``."""
    from anyv2v_amd.pipeline import SourceFeatureCache
    from anyv2v_amd.utils import LatentTrajectory
    c = SourceFeatureCache(max_bytes=3 * 1024)
    f = {"a": torch.zeros(512, dtype=torch.float16)}          # 1 KiB per step
    st = (True, False)
    assert c.store(981, st, f) and c.store(961, st, f) and c.store(941, st, f)
    assert not c.store(921, st, f) and c.skipped_steps == 1 and c.nbytes() == 3 * 1024
    assert c.has(981, st) and not c.has(921, st)
    assert not c.has(981, (True, True))                        # another injection state of the same step: not replayable
    c.steps[(981, st)]["a"].fill_(1)                           # stored tensors are copies
    assert float(f["a"].sum()) == 0.0
    c.bind(("other clip",))
    assert c.nbytes() == 0 and c.store(921, st, f)
    serials = [LatentTrajectory().serial for _ in range(4)]
    assert len(set(serials)) == 4 and serials == sorted(serials)


def test_fused_feed_forward_packing_and_dispatch(monkeypatch):
    """The fused feed-forward contract (AnyV2VFFDesc): ``ops.ff_pack_w2`` + the interleaved GEGLU packing reproduce
    Linear(GEGLU(x)) -- checked against plain torch on the un-packed weights through the op emulation -- and ``FeedForward.run``
    takes the fused op exactly for the 320-channel blocks at the 64x64 level's (hinted) row counts."""
    import torch.nn.functional as F
    from anyv2v_amd import ops
    from anyv2v_amd.unet import FeedForward
    emu.install(monkeypatch)
    torch.manual_seed(3)
    ff = FeedForward(320)
    for prm in ff.parameters():
        prm.data = (torch.randn_like(prm.data.float()) * 0.05).half()
    ff.net[0].pack()
    ff.pack()
    assert tuple(ff._w2s.shape) == (40, 320, 32)
    x, res = (torch.randn(70, 320) * 0.5).half(), torch.randn(70, 320).half()
    proj = x.float() @ ff.net[0].proj.weight.float().t() + ff.net[0].proj.bias.float()
    hid = (proj[:, :1280] * F.gelu(proj[:, 1280:])).half().float()
    ref = (hid @ ff.net[2].weight.float().t() + ff.net[2].bias.float()).half().float() + res.float()
    y = ops.ff_geglu(x, ff.net[0]._w, ff.net[0]._b, ff._w2s, ff.net[2].bias, residual=res)
    assert (y.float() - ref).abs().max() <= 2e-3 * ref.abs().max()
    calls = []
    monkeypatch.setattr(ops, "ff_geglu", lambda *a, **k: calls.append(a[0].shape[0]) or emu.ff_geglu(*a, **k))
    ff.run(x, res)
    assert calls == []                                  # 70 rows: the two GEMMs
    monkeypatch.setattr(ops, "_HINT", [32768, 70])      # as if the launch stood for >= 32768 rows
    ff.run(x, res)
    assert calls == [70]
    wide = FeedForward(640)
    wide.net[0].pack()
    wide.pack()
    assert wide._w2s is None


def test_text_only_checkpoint_folders_get_the_native_text_tower(cpu_ops, tmp_path, monkeypatch):
    """Stable-Diffusion-style checkpoints (ConsistI2V, SEINE's ``sd_path``) have ``text_encoder`` + ``tokenizer`` and no image encoder:
    ``attach_native_text_encoder`` builds the native tower from them (``attach_native_clip_encoders`` would decline and the pipelines
    used to fall back to the synthetic stand-in without a word)."""
    transformers = pytest.importorskip("transformers")
    import types

    from safetensors.torch import save_file

    from anyv2v_amd.encoders import NativeTextEncoder, attach_native_clip_encoders, attach_native_text_encoder
    from hf_clip_reference import HFTextEncoder
    tm, _vm, tcfg, _vcfg = _tiny_clip_pair(transformers)
    d = tmp_path / "text_encoder"
    d.mkdir()
    (d / "config.json").write_text(tcfg.to_json_string())
    save_file({k: v.contiguous() for k, v in tm.state_dict().items()}, str(d / "model.safetensors"))
    holder = types.SimpleNamespace()
    assert not attach_native_text_encoder(holder, str(tmp_path))          # no tokenizer folder yet
    (tmp_path / "tokenizer").mkdir()

    class Tok:
        model_max_length = 16

        def __call__(self, prompts, padding, max_length, truncation, return_tensors):
            ids = torch.zeros(len(prompts), max_length, dtype=torch.long)
            for i, p in enumerate(prompts):
                t = [(ord(c) % 90) + 3 for c in p][: max_length - 1]
                ids[i, : len(t)] = torch.tensor(t, dtype=torch.long)
                ids[i, len(t)] = 2
            return type("O", (), {"input_ids": ids})
    monkeypatch.setattr(transformers.CLIPTokenizer, "from_pretrained", classmethod(lambda cls, path, **kw: Tok()))
    assert attach_native_text_encoder(holder, str(tmp_path)) and isinstance(holder.text_encoder, NativeTextEncoder)
    assert not attach_native_clip_encoders(types.SimpleNamespace(), str(tmp_path))   # (needs the image encoder too)
    dev = torch.device("cpu")
    a = holder.text_encoder.encode(["a robot", ""], dev, clip_skip=None)
    b = HFTextEncoder(tm, Tok()).encode(["a robot", ""], dev, clip_skip=None)
    assert a.shape == b.shape and (a.float() - b.float()).abs().max() <= 8e-3 * float(b.float().abs().max())


def test_several_clips_inverted_in_one_batch(cpu_ops, tmp_path):
    """``pipe.invert_clips``: B clips through ONE forward per step, each row with its own conditioning and latents; the trajectories are
    those of ``pipe.invert`` clip by clip (not bit for bit: other row counts take other launch plans), the files are written per clip."""
    from PIL import Image
    from anyv2v_amd.encoders import attach_synthetic_encoders
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from anyv2v_amd.schedulers import DDIMInverseScheduler
    native, _, ocfg = gc.build_pair("mini", 1234)
    pipe = I2VGenXLPipeline(unet=native, scheduler=DDIMInverseScheduler())
    attach_synthetic_encoders(pipe)
    pipe.text_encoder.dim = pipe.image_encoder.dim = ocfg.cross_attention_dim
    rng = np.random.RandomState(3)
    clips = []
    for k in range(3):
        frames = [Image.fromarray((rng.rand(64, 64, 3) * 255).astype("uint8")) for _ in range(4)]
        lat = pipe.encode_vae_video(frames, pipe.device, height=64, width=64)
        clips.append(dict(prompt="" if k != 1 else "a dog", image=frames[0], latents=lat))
    kw = dict(height=64, width=64, num_frames=4, num_inference_steps=3, target_fps=8)
    singles = [pipe.invert(prompt=c["prompt"], image=c["image"], latents=c["latents"], guidance_scale=1.0, return_trajectory=True, **kw) for c in clips]
    dirs = [str(tmp_path / f"c{k}") for k in range(3)]
    batched = pipe.invert_clips(clips, output_dirs=dirs, **kw)
    assert len(batched) == 3
    for k in range(3):
        assert sorted(batched[k].keys()) == sorted(singles[k].keys())
        for t in singles[k].keys():
            a, b = batched[k][t].float(), singles[k][t].float()
            assert float((a - b).abs().max()) <= 8e-3 * float(b.abs().max()), (k, t)   # two fp16 paths through the mini UNet, 1-3 steps (measured <= 4.4e-3)
        assert sorted(os.listdir(dirs[k])) == sorted(f"ddim_latents_{t}.pt" for t in singles[k].keys())
    # the clips really are different rows
    assert not torch.equal(batched[0][1], batched[1][1])


def test_capture_guard_is_reentrant_and_parks_released_graphs(monkeypatch):
    """``utils.capture_hip_graph`` under nesting / concurrency -- the cyclic collector stays OFF until the outermost capture
    ends (an inner exit must not re-enable it), and graph objects released while a capture records (``release_graphs``: LRU eviction of
    a step engine) are parked, i.e. not destroyed, until no capture is in progress.  ``torch.cuda.graph`` is replaced by a dummy context."""
    import contextlib
    import gc
    import weakref

    import torch
    from anyv2v_amd import utils

    @contextlib.contextmanager
    def fake_graph(g):
        yield
    monkeypatch.setattr(torch.cuda, "graph", fake_graph)

    class G:   # stands for a CUDAGraph: its death is observable
        pass
    assert gc.isenabled()
    g1, g2 = G(), G()
    r1, r2 = weakref.ref(g1), weakref.ref(g2)
    held = {"a": g1}
    with utils.capture_hip_graph(object()):
        assert not gc.isenabled()
        with utils.capture_hip_graph(object()):
            assert not gc.isenabled()
            del g1
            utils.release_graphs(held)          # released DURING a capture: parked
            assert held == {} and r1() is not None
        assert not gc.isenabled(), "the inner capture's exit re-enabled the collector under the outer one"
        assert r1() is not None
    assert gc.isenabled() and r1() is None      # freed when the last capture ended
    lst = [g2]
    del g2
    utils.release_graphs(lst)                   # no capture in progress: dropped at once
    assert lst == [] and r2() is None
