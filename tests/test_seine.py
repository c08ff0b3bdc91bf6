"""SURVEY.md 8(f) F4 -- the SEINE hook family (``anyv2v_amd/seine.py``) on the CPU: native decoder blocks through the op emulation
vs the fixture produced by the REFERENCE's own ``CrossAttnUpBlock3D`` + ``seine/pnp_utils.py`` (``make_golden.py --seine``), and --
where /root/reference exists -- the fixture vs the reference run live."""
import os
import types
import warnings

import pytest
import torch

import cpu_ops_emulation as emu
import seine_spec as spec
from oracle import ref_stubs

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "seine_decoder_hooks.pt")


def _native_blocks():
    from anyv2v_amd import seine as sn
    return sn, {i: spec.fill_weights(sn.CrossAttnUpBlock3D(**spec.block_kwargs(i)), spec.WEIGHT_SEED) for i in spec.BLOCKS}


def _native_call(blk, x, skips, temb, ehs):
    with torch.no_grad():
        return blk(x.half(), tuple(s.half() for s in skips), temb.half(), encoder_hidden_states=ehs.half()).float()


def test_native_seine_decoder_hooks_vs_reference_fixture(monkeypatch):
    emu.install(monkeypatch)
    sn, blocks = _native_blocks()
    fx = torch.load(FIXTURE)
    out = spec.run_cases(blocks, sn, _native_call)
    for i in spec.BLOCKS:
        assert torch.equal(out[f"block{i}_nohook"], out[f"block{i}_hook_t101"])
        for case in ["nohook"] + [f"hook_t{t}" for t in spec.TS_CASES]:
            got, ref = out[f"block{i}_{case}"], fx[f"block{i}_{case}"]
            assert got.shape == ref.shape
            err = float((got - ref).abs().max() / ref.abs().max())
            l2 = float((got - ref).norm() / ref.norm())
            assert err < 4e-3 and l2 < 2.5e-3, (i, case, err, l2)
        a = out[f"block{i}_nohook"]
        assert torch.equal(a[:1], out[f"block{i}_hook_t981"][:1])
        # every hook kind changes the result: temporal only / + cross-attention / + spatial (+ conv in block 1)
        assert float((out[f"block{i}_hook_t301"][1:] - a[1:]).abs().max() / a.abs().max()) > 0.05
        assert float((out[f"block{i}_hook_t501"][1:] - out[f"block{i}_hook_t301"][1:]).abs().max() / a.abs().max()) > 0.05
        assert float((out[f"block{i}_hook_t981"][1:] - out[f"block{i}_hook_t501"][1:]).abs().max() / a.abs().max()) > 0.05


def test_seine_hook_registration_targets():
    """``seine/pnp_utils.py:121-147,195-196,282-294,363-376,448-458``: the schedule lands on attn1 / attn2 / attn_temp of decoder blocks
    4-11, every other transformer block is switched off (empty schedule), ``t`` is written to all of them and to the conv site."""
    sn, blocks = _native_blocks()
    model = types.SimpleNamespace(unet=spec.StubUNet(blocks))
    sn.register_conv_injection(model, [981])
    sn.register_spatial_attention_pnp(model, torch.tensor([981, 961]))
    sn.register_cross_attention_pnp(model, [981])
    sn.register_temp_attention_pnp(model, [981])
    sn.register_time(model, torch.tensor(981))
    up = model.unet.up_blocks
    assert up[1].resnets[1].injection_schedule == frozenset([981]) and up[1].resnets[1].t == 981
    assert up[1].resnets[0].injection_schedule is None
    for res in (1, 2, 3):
        for b in (0, 1, 2):
            blk = up[res].attentions[b].transformer_blocks[0]
            for name in ("attn1", "attn2", "attn_temp"):
                m = getattr(blk, name)
                assert m.t == 981
                assert m.injection_schedule == (frozenset() if (res == 1 and b == 0) else
                                                frozenset([981, 961]) if name == "attn1" else frozenset([981]))
    assert model.unet.down_blocks[2].attentions[1].transformer_blocks[0].attn_temp.t == 981
    assert model.unet.mid_block.attentions[0].transformer_blocks[0].attn2.t == 981


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_seine_fixture_is_what_the_reference_code_produces_and_keys_match():
    warnings.filterwarnings("ignore")
    att, ublocks, res, pnp, Rotary = ref_stubs.load_reference_seine_decoder()
    rot = Rotary(32)
    ref_blocks = {i: spec.fill_weights(ublocks.CrossAttnUpBlock3D(rotary_emb=rot, **spec.block_kwargs(i)), spec.WEIGHT_SEED).eval()
                  for i in spec.BLOCKS}
    sn, nat_blocks = _native_blocks()
    for i in spec.BLOCKS:
        rs, ns = ref_blocks[i].state_dict(), nat_blocks[i].state_dict()
        assert sorted(rs.keys()) == sorted(ns.keys())
        assert all(tuple(rs[k].shape) == tuple(ns[k].shape) for k in rs)

    def call(blk, x, skips, temb, ehs):
        with torch.no_grad():
            return blk(x, skips, temb, encoder_hidden_states=ehs, use_image_num=0)
    out = spec.run_cases(ref_blocks, pnp, call)
    fx = torch.load(FIXTURE)
    for k, v in fx.items():
        if k != "spec":
            assert torch.allclose(out[k], v, rtol=1e-5, atol=1e-5 * float(v.abs().max())), k
    # the native bias table == the reference's RelativePositionBias.forward
    ta = ref_blocks[1].attentions[0].transformer_blocks[0].attn_temp
    nt = nat_blocks[1].attentions[0].transformer_blocks[0].attn_temp
    for n in (4, 16, 40):
        assert torch.allclose(ta.time_rel_pos_bias(n, device="cpu").float(), nt.time_rel_pos_bias.table(n, "cpu"), atol=2e-3)


# ------------------------------------------------------------------------------------------------- the whole UNet
UNET_FIXTURE = os.path.join(HERE, "golden", "seine_unet.pt")


def _native_unet():
    from anyv2v_amd import seine as sn
    return sn, spec.fill_weights(sn.UNet3DConditionModel(**spec.UNET_CFG), spec.WEIGHT_SEED)


def _native_unet_call(u, sample, t, ehs):
    return u(sample.half(), t, encoder_hidden_states=ehs.half()).sample.float()


def test_native_seine_unet_vs_reference_fixture(monkeypatch):
    """``UNet3DConditionModel.forward`` (``seine/models/unet.py:365-513``) un-hooked and under the four hook families."""
    emu.install(monkeypatch)
    sn, unet = _native_unet()
    fx = torch.load(UNET_FIXTURE)
    out = spec.run_unet_cases(unet, sn, _native_unet_call)
    assert torch.equal(out["unet_nohook_t101"], out["unet_hook_t101"])
    for case in ["nohook"] + [f"hook_t{t}" for t in spec.TS_CASES]:
        got, ref = out[f"unet_{case}"], fx[f"unet_{case}"]
        assert got.shape == ref.shape == (spec.B, 4, spec.UNET_F, spec.UNET_H, spec.UNET_W)
        err = float((got - ref).abs().max() / ref.abs().max())
        l2 = float((got - ref).norm() / ref.norm())
        assert err < 8e-3 and l2 < 4e-3, (case, err, l2)
    a = out["unet_nohook"]
    assert torch.equal(a[:1], out["unet_hook_t981"][:1])
    assert float((a[1:] - out["unet_hook_t981"][1:]).abs().max() / a.abs().max()) > 0.05


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="needs /root/reference")
def test_seine_unet_fixture_is_what_the_reference_code_produces_and_keys_match():
    warnings.filterwarnings("ignore")
    att, ublocks, res, pnp, Rotary = ref_stubs.load_reference_seine_decoder(with_unet=True)
    ref = spec.fill_weights(ublocks.unet.UNet3DConditionModel(**spec.UNET_CFG), spec.WEIGHT_SEED).eval()
    sn, nat = _native_unet()
    rs, ns = ref.state_dict(), nat.state_dict()
    assert sorted(rs.keys()) == sorted(ns.keys())
    assert all(tuple(rs[k].shape) == tuple(ns[k].shape) for k in rs)
    nat.load_state_dict(rs, strict=True)

    def call(u, sample, t, ehs):
        with torch.no_grad():
            return u(sample, t, encoder_hidden_states=ehs).sample
    out = spec.run_unet_cases(ref, pnp, call)
    fx = torch.load(UNET_FIXTURE)
    for k, v in fx.items():
        if k != "spec":
            assert torch.allclose(out[k], v, rtol=1e-5, atol=1e-5 * float(v.abs().max())), k
