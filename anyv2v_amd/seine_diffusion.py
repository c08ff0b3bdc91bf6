"""SEINE's ``diffusion/`` package, sampling side, on the HIP step kernel (``seine/diffusion/__init__.py:10-47``, ``respace.py:13-130``,
``gaussian_diffusion.py:147-733``): ``create_diffusion(timestep_respacing, ...)`` -> a ``SpacedDiffusion`` with ``p_sample`` /
``ddim_sample`` / ``ddim_reverse_sample`` and their loops, same names, arguments and return dicts.

The reference's two runners import ``create_diffusion`` and then sample with diffusers schedulers instead (``run_ddim_inversion.py:86``,
``run_pnp_edit.py:74``: "TODO: Use diffusion instead of scheduler"); SEINE's own sampling scripts are what call this package
(``diffusion.ddim_sample_loop(model.forward_with_cfg, z.shape, z, clip_denoised=False, model_kwargs=..., mask=, x_start=, use_concat=)``).
Here every step is ONE launch of ``anyv2v_guided_step[_noise]_f16`` after the model call: the process' per-step scalars (float64 tables on
the host, as the reference keeps them) become the kernel's coefficients --

    x_t = sa x0 + sb eps,   y = c_x0 x0 + c_eps eps (+ sigma n)

* ``p_sample``      : c_x0 = coef1 + coef2 sa, c_eps = coef2 sb, sigma = sqrt(variance)         (posterior mean of (x0, x_t))
* ``ddim_sample``   : c_x0 = sqrt(a_prev), c_eps = sqrt(1 - a_prev - s^2), sigma = s = eta ...  (eq. 12 of the DDIM paper)
* ``ddim_reverse_sample``: c_x0 = sqrt(a_next), c_eps = sqrt(1 - a_next)

with ``clip_denoised`` / ``denoised_fn`` applied to x0 between two launches (x0 out, then the step from the processed x0: the kernel's
"sample" prediction type re-derives eps from it exactly as ``_predict_eps_from_xstart`` does).  Tensors are fp16 (the kernels' type); the
variance noise is drawn from the global RNG in every ``p_sample`` / ``ddim_sample`` call, as in the reference (in fp32, then rounded).

Not built (training side and options SEINE's sampling never switches on): ``training_losses`` / ``calc_bpd_loop`` / the learned-variance
model types, ``cond_fn`` (classifier guidance), ``ModelMeanType.PREVIOUS_X``.
"""
from __future__ import annotations

import enum
import math

import numpy as np
import torch

from . import ops


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """``gaussian_diffusion.py:128-144``."""
    return np.array([min(1 - alpha_bar((i + 1) / num_diffusion_timesteps) / alpha_bar(i / num_diffusion_timesteps), max_beta)
                     for i in range(num_diffusion_timesteps)])


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    """``gaussian_diffusion.py:98-125``: "linear" (Ho et al., scaled to the number of steps) or "squaredcos_cap_v2"."""
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "squaredcos_cap_v2":
        return betas_for_alpha_bar(num_diffusion_timesteps, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def space_timesteps(num_timesteps, section_counts):
    """``respace.py:13-62``: the timesteps of the original process to keep -- "ddimN" (a fixed integer stride giving exactly N), or N steps
    (a comma-separated list: per equally-sized section) spread evenly with rounding."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


class GaussianDiffusion:
    """The tables of ``gaussian_diffusion.py:156-204`` (float64) and the sampling steps on top of them."""

    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type=LossType.MSE):
        if model_mean_type not in (ModelMeanType.EPSILON, ModelMeanType.START_X):
            raise NotImplementedError(f"{model_mean_type}: epsilon / x0 predictions only")
        if model_var_type not in (ModelVarType.FIXED_SMALL, ModelVarType.FIXED_LARGE):
            raise NotImplementedError(f"{model_var_type}: fixed variances only (sampling side; SEINE builds its process with learn_sigma=False)")
        self.model_mean_type, self.model_var_type, self.loss_type = model_mean_type, model_var_type, loss_type
        betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = (np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
                                               if len(self.posterior_variance) > 1 else np.array([]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)

    # ------------------------------------------------------------------ forward process
    def q_sample(self, x_start, t, noise=None):
        """``:218-233``: x_t = sqrt(a_t) x0 + sqrt(1 - a_t) noise (plain torch: once per clip)."""
        if noise is None:
            noise = torch.randn_like(x_start)
        sa = torch.from_numpy(self.sqrt_alphas_cumprod).to(t.device)[t].float().view(-1, *([1] * (x_start.dim() - 1)))
        sb = torch.from_numpy(self.sqrt_one_minus_alphas_cumprod).to(t.device)[t].float().view(-1, *([1] * (x_start.dim() - 1)))
        return (sa * x_start.float() + sb * noise.float()).to(x_start.dtype)

    # ------------------------------------------------------------------ one model call
    def _map_t(self, t):
        return t

    @staticmethod
    def _index(t):
        """The loops give every batch element the same step; a step is one launch with that step's scalars."""
        vals = set(int(v) for v in t.reshape(-1).tolist())
        if len(vals) != 1:
            raise NotImplementedError(f"one timestep per call (got {sorted(vals)}): split the batch")
        return vals.pop()

    def _predict(self, model, x, t, model_kwargs, mask, x_start, use_concat):
        """``p_mean_variance``'s model call (``:278-292``): optional [x, mask, x_start] channel concat, ``.sample`` / tuple outputs."""
        if x.shape[0] != t.shape[0]:
            raise ValueError(f"t has {t.shape[0]} entries for a batch of {x.shape[0]}")
        inp = torch.cat([x, mask.to(x), x_start.to(x)], dim=1) if use_concat else x
        out = model(inp, self._map_t(t), **(model_kwargs or {}))
        out = getattr(out, "sample", out)
        if isinstance(out, tuple):
            out = out[0]
        if out.shape != x.shape:
            raise ValueError(f"model output {tuple(out.shape)} for a sample of {tuple(x.shape)} (learned variances are not supported)")
        return out

    def _step(self, model, x, t, c_of, *, clip_denoised, denoised_fn, cond_fn, model_kwargs, mask, x_start, use_concat, noise_fn=None):
        """Model call + the step.  ``c_of(i)`` -> (c_x0, c_eps, sigma) of step index i."""
        if cond_fn is not None:
            raise NotImplementedError("cond_fn (classifier guidance) is not built")
        i = self._index(t)
        x = x.to(torch.float16).contiguous()
        e = self._predict(model, x, t, model_kwargs, mask, x_start, use_concat).to(torch.float16).contiguous()
        sa, sb = float(self.sqrt_alphas_cumprod[i]), float(self.sqrt_one_minus_alphas_cumprod[i])
        pred = ops.PRED_EPSILON if self.model_mean_type == ModelMeanType.EPSILON else ops.PRED_SAMPLE
        e1 = e.view(1, -1)
        # pred_xstart: the kernel with (c_x0, c_eps) = (1, 0)
        x0 = ops.guided_step(e1, x, (sa, sb, 1.0, 0.0), b_txt=0, prediction=pred)
        if denoised_fn is not None or clip_denoised:
            if denoised_fn is not None:
                x0 = denoised_fn(x0)
            if clip_denoised:
                x0 = x0.clamp(-1, 1)
            x0 = x0.to(torch.float16).contiguous()
            e1, pred = x0.view(1, -1), ops.PRED_SAMPLE       # (eps re-derived from the processed x0, ``_predict_eps_from_xstart``)
        c_x0, c_eps, sigma = c_of(i)
        # the reference draws its noise in EVERY p_sample / ddim_sample call, also where it is multiplied by zero (t == 0, eta == 0):
        # the global RNG advances the same way here
        noise = noise_fn(x) if noise_fn is not None else None
        if sigma == 0.0:
            noise = None
        y = ops.guided_step(e1, x, (sa, sb, c_x0, c_eps), b_txt=0, prediction=pred, noise=noise, sigma=sigma if noise is not None else 0.0)
        return {"sample": y, "pred_xstart": x0}

    # ------------------------------------------------------------------ ancestral sampling
    def _p_coefficients(self, i):
        sa, sb = float(self.sqrt_alphas_cumprod[i]), float(self.sqrt_one_minus_alphas_cumprod[i])
        c1, c2 = float(self.posterior_mean_coef1[i]), float(self.posterior_mean_coef2[i])
        if self.model_var_type == ModelVarType.FIXED_LARGE:
            var = float(np.append(self.posterior_variance[1], self.betas[1:])[i])
        else:
            var = float(np.exp(self.posterior_log_variance_clipped[i]))
        return c1 + c2 * sa, c2 * sb, (math.sqrt(var) if i != 0 else 0.0)        # (no noise when t == 0, ``:434-436``)

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, mask=None, x_start=None,
                 use_concat=False):
        """``:392-439``: x_{t-1} ~ p(. | x_t) -> {"sample", "pred_xstart"}."""
        return self._step(model, x, t, self._p_coefficients, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                          model_kwargs=model_kwargs, mask=mask, x_start=x_start, use_concat=use_concat,
                          noise_fn=self._noise_like)

    @staticmethod
    def _noise_like(x):
        """``th.randn_like(x)`` of an fp32 sample (the global generator of x's device), rounded to the kernels' fp16."""
        return torch.randn(x.shape, dtype=torch.float32, device=x.device).to(torch.float16)

    def _loop(self, step, model, shape, noise, device, progress, **kw):
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else torch.randn(*shape, device=device)
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            try:
                from tqdm.auto import tqdm
                indices = tqdm(indices)
            except ImportError:
                pass
        for i in indices:
            t = torch.tensor([i] * shape[0], device=device)
            with torch.no_grad():
                out = step(model, img, t, **kw)
            yield out
            img = out["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                                  device=None, progress=False, mask=None, x_start=None, use_concat=False):
        """``:492-545``."""
        yield from self._loop(self.p_sample, model, shape, noise, device, progress, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                              cond_fn=cond_fn, model_kwargs=model_kwargs, mask=mask, x_start=x_start, use_concat=use_concat)

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, device=None,
                      progress=False, mask=None, x_start=None, use_concat=False):
        """``:441-490``."""
        final = None
        for final in self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                                    cond_fn=cond_fn, model_kwargs=model_kwargs, device=device, progress=progress,
                                                    mask=mask, x_start=x_start, use_concat=use_concat):
            pass
        return final["sample"]

    # ------------------------------------------------------------------ DDIM
    def _ddim_coefficients(self, i, eta):
        a, a_prev = float(self.alphas_cumprod[i]), float(self.alphas_cumprod_prev[i])
        sigma = eta * math.sqrt((1 - a_prev) / (1 - a)) * math.sqrt(1 - a / a_prev)
        return math.sqrt(a_prev), math.sqrt(max(1 - a_prev - sigma ** 2, 0.0)), (sigma if i != 0 else 0.0)

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0, mask=None,
                    x_start=None, use_concat=False):
        """``:547-600``."""
        return self._step(model, x, t, lambda i: self._ddim_coefficients(i, eta), clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                          cond_fn=cond_fn, model_kwargs=model_kwargs, mask=mask, x_start=x_start, use_concat=use_concat,
                          noise_fn=self._noise_like)

    def ddim_reverse_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        """``:602-638``: x_{t+1} by the reverse ODE (deterministic path only)."""
        assert eta == 0.0, "Reverse ODE only for deterministic path"

        def c_of(i):
            a_next = float(self.alphas_cumprod_next[i])
            return math.sqrt(a_next), math.sqrt(1 - a_next), 0.0
        return self._step(model, x, t, c_of, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs,
                          mask=None, x_start=None, use_concat=False)

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                                     device=None, progress=False, eta=0.0, mask=None, x_start=None, use_concat=False):
        """``:679-732``."""
        yield from self._loop(self.ddim_sample, model, shape, noise, device, progress, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                              cond_fn=cond_fn, model_kwargs=model_kwargs, eta=eta, mask=mask, x_start=x_start, use_concat=use_concat)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, device=None,
                         progress=False, eta=0.0, mask=None, x_start=None, use_concat=False):
        """``:640-677``."""
        final = None
        for final in self.ddim_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                                       cond_fn=cond_fn, model_kwargs=model_kwargs, device=device, progress=progress,
                                                       eta=eta, mask=mask, x_start=x_start, use_concat=use_concat):
            pass
        return final["sample"]

    def training_losses(self, *a, **k):
        raise NotImplementedError("training is outside this build (sampling side only)")


class SpacedDiffusion(GaussianDiffusion):
    """``respace.py:65-113``: the process on a subset of the original timesteps -- betas recomputed so that the kept steps have the
    original cumulative alphas; the model is called with the ORIGINAL timestep numbers (``_WrappedModel``, ``:116-130``)."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.timestep_map = []
        self.original_num_steps = len(kwargs["betas"])
        base = GaussianDiffusion(**kwargs)
        last, new_betas = 1.0, []
        for i, a in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - a / last)
                last = a
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def _map_t(self, t):
        return torch.tensor(self.timestep_map, device=t.device, dtype=t.dtype)[t]


def create_diffusion(timestep_respacing, noise_schedule="linear", use_kl=False, sigma_small=False, predict_xstart=False, learn_sigma=False,
                     rescale_learned_sigmas=False, diffusion_steps=1000):
    """``seine/diffusion/__init__.py:10-47``."""
    if learn_sigma:
        raise NotImplementedError("learn_sigma=True (learned variances): fixed variances only (SEINE's default for its UNet)")
    betas = get_named_beta_schedule(noise_schedule, diffusion_steps)
    loss_type = LossType.RESCALED_KL if use_kl else (LossType.RESCALED_MSE if rescale_learned_sigmas else LossType.MSE)
    if timestep_respacing is None or timestep_respacing == "":
        timestep_respacing = [diffusion_steps]
    return SpacedDiffusion(use_timesteps=space_timesteps(diffusion_steps, timestep_respacing), betas=betas,
                           model_mean_type=ModelMeanType.EPSILON if not predict_xstart else ModelMeanType.START_X,
                           model_var_type=ModelVarType.FIXED_LARGE if not sigma_small else ModelVarType.FIXED_SMALL, loss_type=loss_type)
