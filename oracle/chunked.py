"""Size-safe evaluation of the oracle UNet (TEST ORACLE, not product).

The fp32 / fp16 eager oracles are the CHECKERS of the full-size parity rows.  At BASELINE config 5 with the three-branch PnP
batch ([3,4,128,64,64]) a single eager activation reaches 4.03 G elements (the GEGLU projection, 16 GB in fp32) -- past 2^31
elements and 4 GiB, where vendor kernels with 32-bit offsets are least exercised (the fp32 checker was the
suspected outlier).  ``enable_chunking`` makes every heavy module of an ``I2VGenXLUNetOracle`` process its input in pieces no
larger than what the config-3 rows (validated against the CPU oracle) already use, WITHOUT changing the arithmetic of any output
element:

* per-image modules (``ResnetBlock2D``, ``Transformer2DModel``, up / down samplers, ``conv_in``, ``conv_norm_out``,
  ``conv_out``) run on frame chunks ``[(B f_c), ...]`` -- the batch dimension stays outermost, so the PnP hooks of
  ``oracle.pnp_oracle`` / the reference's ``pnp_utils.py`` (``chunk(3)`` along dim 0) act on each piece exactly as on the whole;
* ``TemporalConvLayer`` (5-D GroupNorm over (C/G, F, H, W) per sample + (3,1,1) convolutions) runs per batch element;
* ``TransformerTemporalModel`` (``unet_oracle.py:196-209``): its 5-D GroupNorm per batch element, then projection / blocks /
  projection on row chunks of the pixel grid ``[(B h_c W), F, C]`` (temporal attention is independent per pixel; hooks again see
  the batch outermost).

``tests/test_oracle.py::test_chunked_oracle_equals_plain_oracle`` checks chunked == plain (hooked and un-hooked) on the CPU.
Call order: register hooks -> ``enable_chunking`` -> forward(s) -> ``disable_chunking`` -> clear hooks.
"""
from __future__ import annotations

import torch

_SAVED = "_chunked_saved_forward"


def _set_forward(mod, fn):
    mod.__dict__[_SAVED] = mod.__dict__.get("forward", None)   # an instance-level forward = a PnP hook's replacement
    mod.__dict__["forward"] = fn


def _frame_chunked(mod, frame_chunk):
    inner = mod.forward   # bound method: the class forward, or the hook's replacement

    def forward(*args, **kw):
        st = mod._chunk_state
        B, Fr = st["B"], st["F"]
        n = B * Fr
        if Fr <= frame_chunk:
            return inner(*args, **kw)
        outs = []
        for f0 in range(0, Fr, frame_chunk):
            f1 = min(Fr, f0 + frame_chunk)

            def cut(a):
                if torch.is_tensor(a) and a.dim() >= 1 and a.shape[0] == n:
                    return a.reshape((B, Fr) + tuple(a.shape[1:]))[:, f0:f1].reshape((B * (f1 - f0),) + tuple(a.shape[1:]))
                return a

            o = inner(*[cut(a) for a in args], **{k: cut(v) for k, v in kw.items()})
            outs.append(o.reshape((B, f1 - f0) + tuple(o.shape[1:])))
        o = torch.cat(outs, dim=1)
        return o.reshape((n,) + tuple(o.shape[2:]))

    return forward


def _batch_chunked_tconv(mod):
    inner = mod.forward

    def forward(x, num_frames):
        B = x.shape[0] // num_frames
        if B == 1:
            return inner(x, num_frames)
        return torch.cat([inner(x[i * num_frames:(i + 1) * num_frames], num_frames) for i in range(B)], dim=0)

    return forward


def _pixel_chunked_temporal(mod, row_chunk):
    def forward(x, num_frames):
        bf, c, h, w = x.shape
        b = bf // num_frames
        res = x
        xn = torch.cat([mod.norm(x[i * num_frames:(i + 1) * num_frames][None].permute(0, 2, 1, 3, 4)) for i in range(b)], dim=0)
        out = torch.empty((b, num_frames, c, h, w), dtype=x.dtype, device=x.device)
        for r0 in range(0, h, row_chunk):
            r1 = min(h, r0 + row_chunk)
            xs = xn[:, :, :, r0:r1].permute(0, 3, 4, 2, 1).reshape(b * (r1 - r0) * w, num_frames, c)
            xs = mod.proj_in(xs)
            for blk in mod.transformer_blocks:
                xs = blk(xs, None)
            xs = mod.proj_out(xs)
            out[:, :, :, r0:r1] = xs.reshape(b, r1 - r0, w, num_frames, c).permute(0, 3, 4, 1, 2)
        return out.reshape(bf, c, h, w) + res

    return forward


def _modules(unet):
    spatial, tconv, ttrans = [unet.conv_in, unet.conv_norm_out, unet.conv_out], [], [unet.transformer_in]
    blocks = list(unet.down_blocks) + [unet.mid_block] + list(unet.up_blocks)
    for blk in blocks:
        spatial += list(blk.resnets)
        tconv += list(blk.temp_convs)
        if getattr(blk, "attentions", None) is not None:
            spatial += list(blk.attentions)
            ttrans += list(blk.temp_attentions)
        for name in ("downsamplers", "upsamplers"):
            if getattr(blk, name, None) is not None:
                spatial += list(getattr(blk, name))
    return spatial, tconv, ttrans


def enable_chunking(unet: "uo.I2VGenXLUNetOracle", batch: int, num_frames: int, frame_chunk: int = 16, row_chunk: int = 8):
    """Install the chunked forwards (instance attributes; hooks registered BEFORE this call stay in effect inside the pieces)."""
    spatial, tconv, ttrans = _modules(unet)
    state = {"B": batch, "F": num_frames}
    for m in spatial:
        m._chunk_state = state
        _set_forward(m, _frame_chunked(m, frame_chunk))
    for m in tconv:
        _set_forward(m, _batch_chunked_tconv(m))
    for m in ttrans:
        _set_forward(m, _pixel_chunked_temporal(m, row_chunk))
    unet._chunked = True


def disable_chunking(unet):
    spatial, tconv, ttrans = _modules(unet)
    for m in spatial + tconv + ttrans:
        if _SAVED in m.__dict__:
            prev = m.__dict__.pop(_SAVED)
            if prev is None:
                m.__dict__.pop("forward", None)
            else:
                m.__dict__["forward"] = prev
        m.__dict__.pop("_chunk_state", None)
    unet._chunked = False
