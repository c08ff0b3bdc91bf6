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
    """(conv_on, spatial_on, temporal_on) for the currently registered time -- keys the HIP-graph cache."""
    from .unet import pnp_on
    unet = model.unet
    r = unet.up_blocks[1].resnets[1]
    sp = unet.up_blocks[2].attentions[0].transformer_blocks[0].attn1.processor
    tp = unet.up_blocks[2].temp_attentions[0].transformer_blocks[0].attn1.processor
    return (pnp_on(r.t, r.injection_schedule), pnp_on(getattr(sp, "t", None), getattr(sp, "injection_schedule", None)),
            pnp_on(getattr(tp, "t", None), getattr(tp, "injection_schedule", None)))
