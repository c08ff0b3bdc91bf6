"""Plug-and-play feature injection: same four entry points as the reference's ``i2vgen-xl/pnp_utils.py``
(``register_time`` :19, ``register_conv_injection`` :39, ``register_spatial_attention_pnp`` :140,
``register_temp_attention_pnp`` :246), same call signatures (``model`` is the pipeline, ``model.unet`` the UNet),
same sites, same schedule semantics -- but the hooks are native: they only set ``.t`` / ``.injection_schedule``
on the HIP-backed modules, and the injection itself happens inside the kernels' addressing
(``HipAttnProcessor`` qk_mod aliasing, ``ResnetBlock2D.run`` source-only main path).

Differences from the reference, all exact:
  * schedules are stored as frozensets of python ints, so the per-step ``t in schedule`` test does not
    touch the device (the reference's ``t in tensor`` + ``t.item()`` sync the GPU every step);
  * no tensor copies (reference: 4 slice copies per attention site, 2 per conv site).
"""
from __future__ import annotations

import logging

from .unet import HipAttnProcessor

logger = logging.getLogger(__name__)

_TIME_SITES = {1: [0, 1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}   # pnp_utils.py:22
_ATTN_SITES = {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}      # pnp_utils.py:235,340


def _as_schedule(injection_schedule):
    if injection_schedule is None:
        return None
    if hasattr(injection_schedule, "tolist"):
        injection_schedule = injection_schedule.tolist()
    return frozenset(int(t) for t in injection_schedule)


def register_time(model, t):
    t = int(t)
    unet = model.unet
    setattr(unet.up_blocks[1].resnets[1], "t", t)
    for res, blocks in _TIME_SITES.items():
        for block in blocks:
            setattr(unet.up_blocks[res].attentions[block].transformer_blocks[0].attn1.processor, "t", t)
            setattr(unet.up_blocks[res].temp_attentions[block].transformer_blocks[0].attn1.processor, "t", t)


def clear_time(model):
    """Forget the registered timestep (hooks inert): needed when inversion (B=1, no hooks in the reference's
    stage-1 process) and PnP editing share one process / one UNet object."""
    unet = model.unet
    setattr(unet.up_blocks[1].resnets[1], "t", None)
    for res, blocks in _TIME_SITES.items():
        for block in blocks:
            setattr(unet.up_blocks[res].attentions[block].transformer_blocks[0].attn1.processor, "t", None)
            setattr(unet.up_blocks[res].temp_attentions[block].transformer_blocks[0].attn1.processor, "t", None)


def register_conv_injection(model, injection_schedule):
    conv_module = model.unet.up_blocks[1].resnets[1]
    setattr(conv_module, "injection_schedule", _as_schedule(injection_schedule))


def register_spatial_attention_pnp(model, injection_schedule):
    sched = _as_schedule(injection_schedule)
    for res, blocks in _ATTN_SITES.items():
        for block in blocks:
            module = model.unet.up_blocks[res].attentions[block].transformer_blocks[0].attn1
            module.processor = HipAttnProcessor(sched)


def register_temp_attention_pnp(model, injection_schedule):
    sched = _as_schedule(injection_schedule)
    for res, blocks in _ATTN_SITES.items():
        for block in blocks:
            module = model.unet.up_blocks[res].temp_attentions[block].transformer_blocks[0].attn1
            module.processor = HipAttnProcessor(sched)


def injection_state(model):
    """Injection decision of every hook site for the currently registered time: (conv, 8 x spatial, 8 x temporal).  It
    keys the HIP-graph cache (a captured step bakes each site's decision in), so it is built from ALL 17 sites -- a
    single site with its own schedule or processor (the reference's commented "Disable PNP" per-module pattern,
    ``pnp_utils.py:229-232``) gets its own graph instead of silently replaying another state's."""
    from .unet import pnp_on
    unet = model.unet
    r = unet.up_blocks[1].resnets[1]
    state = [pnp_on(r.t, r.injection_schedule)]
    for kind in ("attentions", "temp_attentions"):
        for res, blocks in _ATTN_SITES.items():
            for block in blocks:
                p = getattr(unet.up_blocks[res], kind)[block].transformer_blocks[0].attn1.processor
                state.append(pnp_on(getattr(p, "t", None), getattr(p, "injection_schedule", None)))
    return tuple(state)


def injection_sites(model):
    """The 17 hook sites in ``injection_state`` order: [(name, object carrying .t / .injection_schedule / .src_io, columns)], columns
    = width of the source features a multi-edit job keeps per site (conv: Cout; attention: Q | K = 2 x inner dim)."""
    unet = model.unet
    r = unet.up_blocks[1].resnets[1]
    sites = [("conv.up1.res1", r, r.out_channels)]
    for kind in ("attentions", "temp_attentions"):
        for res, blocks in _ATTN_SITES.items():
            for block in blocks:
                a = getattr(unet.up_blocks[res], kind)[block].transformer_blocks[0].attn1
                sites.append((f"{kind}.up{res}.{block}", a.processor, 2 * a.inner_dim))
    return sites


def has_foreign_hooks(unet) -> bool:
    """True when a torch-style processor / ``forward`` from outside this package is plugged into the seams B1 / B2 (e.g.
    the reference's own ``pnp_utils.py`` hooks): such code may sync or allocate, so steps are not captured into HIP graphs."""
    from .unet import Attention, ResnetBlock2D
    for m in unet.modules():
        if isinstance(m, Attention) and not isinstance(m.processor, HipAttnProcessor):
            return True
        if isinstance(m, ResnetBlock2D) and "forward" in m.__dict__:
            return True
    return False
