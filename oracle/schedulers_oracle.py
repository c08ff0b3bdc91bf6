"""DDIM / inverse-DDIM scheduler restatement (TEST ORACLE, not product).

Follows the vendored ``/root/reference/consisti2v/ddim_inverse_scheduler.py``:
betas ``:49-90`` (squaredcos_cap_v2), zero-terminal-SNR rescale ``:94-127``,
timesteps ``:253-289``, inverse step ``:329-369``.  The forward ``DDIMScheduler``
(diffusers 0.26.3, not vendored) is the mirror image -- and is pinned to the reference's own
``seine/diffusion/gaussian_diffusion.py::ddim_sample`` on the respaced timesteps (``tests/test_oracle.py``).  Config in
effect is the one logged at ``i2vgen-xl/demo.ipynb:1208-1226``.

Deliberately written with float64 numpy + explicit loops so it shares no code with
``anyv2v_amd.schedulers``.
"""
from __future__ import annotations

import math

import numpy as np

NUM_TRAIN = 1000


def alphas_cumprod() -> np.ndarray:
    """float32 table exactly as the reference builds it (torch fp32 semantics emulated in numpy)."""
    def abar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = np.array([min(1 - abar((i + 1) / NUM_TRAIN) / abar(i / NUM_TRAIN), 0.999) for i in range(NUM_TRAIN)],
                     dtype=np.float32)
    # rescale_zero_terminal_snr (fp32 arithmetic like torch)
    alphas = (np.float32(1.0) - betas).astype(np.float32)
    ac = np.cumprod(alphas, dtype=np.float32)
    s = np.sqrt(ac).astype(np.float32)
    s0, sT = s[0].copy(), s[-1].copy()
    s = (s - sT).astype(np.float32)
    s = (s * (s0 / (s0 - sT))).astype(np.float32)
    ab = (s ** 2).astype(np.float32)
    al = np.concatenate([ab[0:1], (ab[1:] / ab[:-1]).astype(np.float32)])
    betas = (np.float32(1.0) - al).astype(np.float32)
    return np.cumprod((np.float32(1.0) - betas).astype(np.float32), dtype=np.float32)


def ddim_timesteps(n: int) -> np.ndarray:
    r = NUM_TRAIN // n
    return (np.arange(n) * r).round()[::-1].astype(np.int64) + 1


def inverse_timesteps(n: int) -> np.ndarray:
    r = NUM_TRAIN // n
    return (np.arange(n) * r).round().astype(np.int64) + 1


def ddim_step(v: np.ndarray, t: int, x: np.ndarray, n: int, ac: np.ndarray) -> np.ndarray:
    """eta=0, v-prediction, no clipping, set_alpha_to_one."""
    p = t - NUM_TRAIN // n
    a_t = float(ac[t])
    a_p = float(ac[p]) if p >= 0 else 1.0
    x0 = math.sqrt(a_t) * x - math.sqrt(1 - a_t) * v
    eps = math.sqrt(a_t) * v + math.sqrt(1 - a_t) * x
    return math.sqrt(a_p) * x0 + math.sqrt(1 - a_p) * eps


def inverse_step(v: np.ndarray, t: int, x: np.ndarray, n: int, ac: np.ndarray) -> np.ndarray:
    c = min(t - NUM_TRAIN // n, NUM_TRAIN - 1)
    a_c = float(ac[c]) if c >= 0 else 1.0
    a_n = float(ac[t])
    x0 = math.sqrt(a_c) * x - math.sqrt(1 - a_c) * v
    eps = math.sqrt(a_c) * v + math.sqrt(1 - a_c) * x
    return math.sqrt(a_n) * x0 + math.sqrt(1 - a_n) * eps
