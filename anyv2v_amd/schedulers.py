"""DDIM and inverse-DDIM schedulers with the diffusers call surface the reference relies on (seam B4):
``set_timesteps(n, device=)``, assignable ``.timesteps``, ``scale_model_input``, ``step(...).prev_sample``,
``init_noise_sigma``, ``order``, ``from_pretrained(repo, subfolder="scheduler")``
(used at ``i2vgen-xl/run_group_ddim_inversion.py:92-100`` and ``pipeline_i2vgen_xl.py:812,868,1104,1173,1359,1418``).

Numerics follow the vendored inverse scheduler ``consisti2v/ddim_inverse_scheduler.py`` (betas :72-90,
zero-terminal-SNR rescale :94-127, timesteps :253-289, step :329-369) with the configuration the reference run
logged at ``i2vgen-xl/demo.ipynb:1208-1226``.  The elementwise update itself runs in the HIP kernels
(``anyv2v_ddim_step_f16``; or fused with CFG in ``anyv2v_cfg_ddim_step_f16`` on the fast path).
"""
from __future__ import annotations

import json
import math
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import ops

I2VGEN_XL_SCHEDULER_CONFIG = dict(  # i2vgen-xl/demo.ipynb:1208-1226
    num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="squaredcos_cap_v2",
    trained_betas=None, clip_sample=False, clip_sample_range=1.0, set_alpha_to_one=True, steps_offset=1,
    prediction_type="v_prediction", thresholding=False, dynamic_thresholding_ratio=0.995, sample_max_value=1.0,
    timestep_spacing="leading", rescale_betas_zero_snr=True)


# The ConsistI2V release's ``scheduler/scheduler_config.json`` is not in the reference tree (``run_pnp_edit.py:58-61`` downloads it).
# What the tree pins: ``configs/pipeline_256/pnp_edit.yaml:27`` quotes timesteps 981 / 921 / 801 / 581 for indices 0 / 3 / 9 / 20 of
# 50 steps = "leading" spacing with steps_offset 1.  The rest is the Stable-Diffusion-1.x noise schedule ConsistI2V was trained on
# (linear betas 0.00085 .. 0.012, epsilon prediction, no terminal-SNR rescale); a local ``scheduler_config.json`` overrides it.
CONSISTI2V_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", trained_betas=None, clip_sample=False,
    clip_sample_range=1.0, set_alpha_to_one=True, steps_offset=1, prediction_type="epsilon", thresholding=False,
    dynamic_thresholding_ratio=0.995, sample_max_value=1.0, timestep_spacing="leading", rescale_betas_zero_snr=False)

# SEINE builds its schedulers from Stable Diffusion 1.4's ``scheduler/scheduler_config.json`` with the betas overridden by its own
# config (``seine/run_pnp_edit.py:86-103``, ``configs/pnp_edit.yaml:28-31``: linear 1e-4 .. 0.02).  The SD-1.4 file is not in the
# reference tree; these are its fields (epsilon prediction, ``set_alpha_to_one`` false, ``steps_offset`` 1, no sample clipping).  The
# reference's ``load_ddim_latents_at_t(t + 1)`` under the DDPM sampler (``run_pnp_edit.py:170``) pins "DDPM timesteps = DDIM
# timesteps - 1", i.e. diffusers-0.15's ``DDPMScheduler.set_timesteps`` without ``steps_offset`` (``seine/requirement.txt:5``).
SEINE_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None, clip_sample=False,
    clip_sample_range=1.0, set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon", thresholding=False,
    dynamic_thresholding_ratio=0.995, sample_max_value=1.0, timestep_spacing="leading", rescale_betas_zero_snr=False)

_PREDICTION = {"v_prediction": ops.PRED_V, "epsilon": ops.PRED_EPSILON, "sample": ops.PRED_SAMPLE}


def _cosine_betas(n: int, max_beta: float = 0.999) -> torch.Tensor:
    t = np.arange(n + 1, dtype=np.float64) / n
    abar = np.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return torch.tensor(np.minimum(1.0 - abar[1:] / abar[:-1], max_beta), dtype=torch.float32)


def _zero_terminal_snr(betas: torch.Tensor) -> torch.Tensor:
    s = torch.cumprod(1.0 - betas, 0).sqrt()
    s0, sT = s[0].clone(), s[-1].clone()
    s = (s - sT) * (s0 / (s0 - sT))
    abar = s ** 2
    alphas = torch.cat([abar[:1], abar[1:] / abar[:-1]])
    return 1.0 - alphas


class SchedulerOutput:
    def __init__(self, prev_sample, pred_original_sample=None):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class _DDIMBase:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, **kwargs):
        cfg = dict(I2VGEN_XL_SCHEDULER_CONFIG)
        cfg.update(kwargs)
        self.config = SimpleNamespace(**cfg)
        n = cfg["num_train_timesteps"]
        if cfg["trained_betas"] is not None:
            betas = torch.tensor(cfg["trained_betas"], dtype=torch.float32)
        elif cfg["beta_schedule"] == "linear":
            betas = torch.linspace(cfg["beta_start"], cfg["beta_end"], n, dtype=torch.float32)
        elif cfg["beta_schedule"] == "scaled_linear":
            betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, n, dtype=torch.float32) ** 2
        elif cfg["beta_schedule"] == "squaredcos_cap_v2":
            betas = _cosine_betas(n)
        else:
            raise NotImplementedError(f"{cfg['beta_schedule']} is not implemented for {self.__class__}")
        if cfg["rescale_betas_zero_snr"]:
            betas = _zero_terminal_snr(betas)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg["set_alpha_to_one"] else self.alphas_cumprod[0]
        self.initial_alpha_cumprod = self.final_alpha_cumprod
        self.num_inference_steps = None
        self.timesteps = torch.arange(n - 1, -1, -1, dtype=torch.int64)
        if cfg["prediction_type"] not in _PREDICTION or cfg["clip_sample"] or cfg["thresholding"]:
            raise NotImplementedError("prediction_type in ('v_prediction', 'epsilon', 'sample') without clipping / thresholding only")
        self.prediction = _PREDICTION[cfg["prediction_type"]]

    @classmethod
    def from_pretrained(cls, repo, subfolder=None, **kw):
        """Offline loader: a local directory with ``scheduler_config.json`` if it exists, else the I2VGen-XL config."""
        path = os.path.join(repo, subfolder or "", "scheduler_config.json")
        cfg = {}
        if os.path.isfile(path):
            with open(path) as f:
                cfg = {k: v for k, v in json.load(f).items() if k in I2VGEN_XL_SCHEDULER_CONFIG}
        cfg.update(kw)
        return cls(**cfg)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        """``sqrt(a_t) x + sqrt(1 - a_t) noise`` per batch element (diffusers' ``add_noise``; the FrameInit path of the ConsistI2V
        pipeline, ``pipeline_video_editing.py:622-631``).  Once per clip, on whatever device the tensors live."""
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=torch.float32)[timesteps.to(original_samples.device).long()]
        shape = (-1,) + (1,) * (original_samples.dim() - 1)
        return (ac.sqrt().view(shape) * original_samples.float() + (1 - ac).sqrt().view(shape) * noise.float()).to(original_samples.dtype)

    def _ratio(self) -> int:
        return self.config.num_train_timesteps // self.num_inference_steps

    def _check_n(self, n):
        if n > self.config.num_train_timesteps:
            raise ValueError(f"`num_inference_steps`: {n} cannot be larger than `self.config.train_timesteps`:"
                             f" {self.config.num_train_timesteps}")
        if self.config.timestep_spacing != "leading":
            raise NotImplementedError("only timestep_spacing='leading' is implemented")

    def _abar(self, t: int) -> float:
        return float(self.alphas_cumprod[t]) if t >= 0 else float(self.final_alpha_cumprod)

    def coefficients(self, timestep: int):
        raise NotImplementedError

    def eta_coefficients(self, timestep: int, eta: float):
        raise NotImplementedError(f"eta > 0 is the forward DDIM scheduler's option, not {type(self).__name__}'s")

    def coefficient_table(self, timesteps, device) -> torch.Tensor:
        """[len(timesteps), 4] fp32 device table {sqrt(a_t), sqrt(1-a_t), sqrt(a_p), sqrt(1-a_p)} for the fused step."""
        rows = [self.coefficients(int(t)) for t in timesteps]
        return torch.tensor(rows, dtype=torch.float32, device=device)

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if eta != 0.0:
            # stochastic DDIM (Song et al. eq. 12/16; unused by AnyV2V's runners): x' = sqrt(a_p) x0 + sqrt(1 - a_p - s^2) eps + s n
            sa_t, sb_t, cx, ce, sigma = self.eta_coefficients(int(timestep), eta)
            e = model_output.to(torch.float16).contiguous()
            noise = variance_noise if variance_noise is not None else self.draw_noise(e, generator)
            prev = ops.guided_step(e.view(1, -1), sample.to(torch.float16).contiguous(), (sa_t, sb_t, cx, ce), b_txt=0,
                                   prediction=self.prediction, noise=noise.to(e).contiguous(), sigma=sigma)
            return SchedulerOutput(prev) if return_dict else (prev,)
        sa_t, sb_t, sa_p, sb_p = self.coefficients(int(timestep))
        if self.prediction == ops.PRED_V:
            prev = ops.ddim_step(model_output, sample, sa_t, sb_t, sa_p, sb_p)
        else:
            x = sample.to(torch.float16).contiguous()
            prev = ops.guided_step(model_output.to(torch.float16).contiguous().view(1, -1), x, (sa_t, sb_t, sa_p, sb_p), b_txt=0,
                                   prediction=self.prediction)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev)

    def __len__(self):
        return self.config.num_train_timesteps


class DDIMScheduler(_DDIMBase):
    def set_timesteps(self, num_inference_steps: int, device=None):
        self._check_n(num_inference_steps)
        self.num_inference_steps = num_inference_steps
        r = self._ratio()
        ts = (np.arange(0, num_inference_steps) * r).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device)

    def coefficients(self, timestep: int):
        prev = timestep - self._ratio()
        a_t, a_p = self._abar(timestep), self._abar(prev)
        return (math.sqrt(a_t), math.sqrt(1.0 - a_t), math.sqrt(a_p), math.sqrt(1.0 - a_p))

    def eta_coefficients(self, timestep: int, eta: float):
        """(sa_t, sb_t, c_x0, c_eps, sigma) of ``ops.guided_step(..., noise=, sigma=)`` for ``eta > 0``: sigma = eta sqrt((1 - a_p) /
        (1 - a_t)) sqrt(1 - a_t / a_p), c_eps = sqrt(1 - a_p - sigma^2) (pinned to ``seine/diffusion/gaussian_diffusion.py:583-599``)."""
        prev = timestep - self._ratio()
        a_t, a_p = self._abar(timestep), self._abar(prev)
        sigma = float(eta) * math.sqrt((1.0 - a_p) / (1.0 - a_t)) * math.sqrt(max(1.0 - a_t / a_p, 0.0))
        return (math.sqrt(a_t), math.sqrt(1.0 - a_t), math.sqrt(a_p), math.sqrt(max(1.0 - a_p - sigma * sigma, 0.0)), sigma)

    noise_on_host = False   # eta > 0: the variance noise is drawn where diffusers' randn_tensor draws it (on the sample's device, or on the
                            # host when the generator is a CPU generator); True forces the host draw (a seed then means the same sample on any device)

    def draw_noise(self, like: torch.Tensor, generator=None):
        if self.noise_on_host or (generator is not None and generator.device.type == "cpu"):
            return torch.randn(like.shape, generator=generator, dtype=torch.float32).to(device=like.device, dtype=like.dtype)
        return torch.randn(like.shape, generator=generator, device=like.device, dtype=like.dtype)


class DDIMInverseScheduler(_DDIMBase):
    def set_timesteps(self, num_inference_steps: int, device=None):
        self._check_n(num_inference_steps)
        self.num_inference_steps = num_inference_steps
        r = self._ratio()
        ts = (np.arange(0, num_inference_steps) * r).round().copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device)

    def step(self, model_output, timestep, sample, return_dict: bool = True):
        """The vendored scheduler's surface (``consisti2v/ddim_inverse_scheduler.py:291-297``): no ``eta`` / ``generator`` -- a pipeline's
        ``prepare_extra_step_kwargs`` therefore hands it neither."""
        return super().step(model_output, timestep, sample, return_dict=return_dict)

    def coefficients(self, timestep: int):
        # the sample lives at level c = t - r (alpha 1.0 below 0); the step takes it to level t
        cur = min(timestep - self._ratio(), self.config.num_train_timesteps - 1)
        a_c, a_n = self._abar(cur), self._abar(timestep)
        return (math.sqrt(a_c), math.sqrt(1.0 - a_c), math.sqrt(a_n), math.sqrt(1.0 - a_n))


class DDPMScheduler(_DDIMBase):
    """diffusers-0.15 ``DDPMScheduler`` as SEINE's edit loop calls it (``seine/run_pnp_edit.py:93-103,205``: ``set_timesteps``,
    ``timesteps``, ``step(noise_pred, t, x)["prev_sample"]``; variance type "fixed_small", no sample clipping): the ancestral step
    ``x' = c0 x0 + ct x + sqrt(var) n`` with ``c0 = sqrt(a_prev) beta_t / (1 - a_t)``, ``ct = sqrt(alpha_t) (1 - a_prev) / (1 - a_t)``,
    ``var = (1 - a_prev) / (1 - a_t) beta_t`` (0 at t = 0), ``alpha_t = a_t / a_prev``, ``a_prev`` = 1 below timestep 0.  The noise is
    drawn from the global RNG (or ``generator``) in the sample's dtype on the sample's device, as ``randn_tensor`` does.  diffusers is
    not in the reference tree (restated from the 0.15.0 release the reference pins, ``seine/requirement.txt:5``), but the process is:
    the coefficients are pinned to the posterior mean / FIXED_SMALL variance of ``seine/diffusion/gaussian_diffusion.py`` on the
    respaced timesteps (``tests/test_seine.py::test_ddpm_step_is_pinned_to_the_references_own_gaussian_diffusion``)."""

    def set_timesteps(self, num_inference_steps: int, device=None):
        self._check_n(num_inference_steps)
        self.num_inference_steps = num_inference_steps
        ts = (np.arange(0, num_inference_steps) * self._ratio()).round()[::-1].copy().astype(np.int64)   # (no steps_offset in 0.15)
        self.timesteps = torch.from_numpy(ts).to(device)

    def ancestral_coefficients(self, timestep: int):
        """(sa_t, sb_t, c_x0, c_eps, sigma) of ``ops.guided_step``: x = sa_t x0 + sb_t eps, so c0 x0 + ct x = (c0 + ct sa_t) x0 + ct sb_t eps."""
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        prev = timestep - self.config.num_train_timesteps // n
        a_t = float(self.alphas_cumprod[timestep])
        a_p = float(self.alphas_cumprod[prev]) if prev >= 0 else 1.0
        alpha_t = a_t / a_p
        beta_t = 1.0 - alpha_t
        c0 = math.sqrt(a_p) * beta_t / (1.0 - a_t)
        ct = math.sqrt(alpha_t) * (1.0 - a_p) / (1.0 - a_t)
        var = max((1.0 - a_p) / (1.0 - a_t) * beta_t, 1e-20)
        sa_t, sb_t = math.sqrt(a_t), math.sqrt(1.0 - a_t)
        return sa_t, sb_t, c0 + ct * sa_t, ct * sb_t, (math.sqrt(var) if timestep > 0 else 0.0)

    def coefficients(self, timestep: int):
        return self.ancestral_coefficients(timestep)[:4]

    noise_on_host = False   # tests: draw from the CPU generator (the fixture's stream) whatever device the sample lives on

    def draw_noise(self, like: torch.Tensor, timestep: int, generator=None):
        if int(timestep) <= 0:
            return None
        if self.noise_on_host:
            return torch.randn(like.shape, generator=generator, dtype=like.dtype).to(like.device)
        return torch.randn(like.shape, generator=generator, device=like.device, dtype=like.dtype)

    def step(self, model_output, timestep, sample, generator=None, return_dict: bool = True, **unused):
        if self.prediction != ops.PRED_EPSILON or self.config.clip_sample:
            raise NotImplementedError("DDPM step: epsilon prediction without sample clipping only")
        sa_t, sb_t, cx, ce, sigma = self.ancestral_coefficients(int(timestep))
        e = model_output.to(torch.float16).contiguous()
        x = sample.to(torch.float16).contiguous()
        noise = self.draw_noise(e, int(timestep), generator)
        prev = ops.guided_step(e.view(1, -1), x, (sa_t, sb_t, cx, ce), b_txt=0, prediction=self.prediction, noise=noise, sigma=sigma)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev)

