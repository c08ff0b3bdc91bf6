"""Run the REFERENCE's own SEINE runner classes on the CPU (TEST INFRASTRUCTURE; needs /root/reference, this container only).

``/root/reference/seine/run_ddim_inversion.py`` and ``run_pnp_edit.py`` are imported verbatim: ``SEINEDDIMInversionPipeline``
(``ddim_inversion`` ``:127-168``, ``ddim_sample`` ``:171-199``, ``extract_ddim_latents`` ``:202-273``, ``decode_latents``) and
``SEINEPnPPipeline`` (``get_ddim_latents_path``, ``get_ddim_inversion_prompt``, ``init_pnp`` ``:209-243``,
``compute_masked_video_latents_at_0``, ``edit_video`` ``:265-343``, ``sample_loop``, ``denoise_step`` ``:162-207``) then run unmodified
around the reference's own ``UNet3DConditionModel`` (``oracle.ref_stubs.load_reference_seine_decoder(with_unet=True)``), its own
``seine/pnp_utils.py`` (hooks, ``load_video_frames``, ``load_ddim_latents_at_*``), ``seine_utils.mask_generation_before`` and
``datasets/video_transforms.py``.  The two ``__init__`` methods cannot run (they download Stable Diffusion 1.4 and ``seine.pt``): the
objects are allocated without them and given the attributes those constructors set, computed by the same statements.

Stand-ins for what is absent here: torchvision's ``Compose`` / ``Normalize`` (as in ``ref_consisti2v_pipeline``), OmegaConf (the
product's ``anyv2v_amd.config`` subset: attribute access and ``to_container`` only), diffusers' ``DDIMScheduler`` (``alphas_cumprod``
from the vendored inverse scheduler's table; the runners do their own DDIM arithmetic and only read ``timesteps`` /
``alphas_cumprod`` / ``final_alpha_cumprod``) and ``DDPMScheduler`` (``RefDDPM``: the diffusers-0.15.0 ancestral step restated in
fp64 -- UNPINNED, diffusers is not in the tree), VAE / text encoder: the deterministic toys of ``oracle.ref_pipeline``.  The
reference casts its inputs to fp16 (``.to(dtype=torch.float16)``) for fp16 GPU modules; the fp32 CPU modules here sit behind
wrappers that take those fp16 tensors, compute in fp32 and -- the UNet -- return fp16 like the real module, so every fp16 rounding
point of the reference's data flow is kept.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch
from torch import nn

from . import ref_consisti2v_pipeline as rcp
from . import ref_pipeline as rp
from . import ref_stubs

SCHED_CFG = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", clip_sample=False,
                 set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon", timestep_spacing="leading",
                 rescale_betas_zero_snr=False)


# ----------------------------------------------------------------------------------------------- schedulers
class RefDDIM:
    """What the runners read of diffusers' ``DDIMScheduler``: ``set_timesteps`` / ``timesteps`` ("leading", steps_offset 1),
    ``alphas_cumprod``, ``final_alpha_cumprod`` (``set_alpha_to_one`` false: the first table entry) and ``step`` (eta 0)."""

    def __init__(self):
        inv = ref_stubs.load_reference_inverse_scheduler().DDIMInverseScheduler(**SCHED_CFG)
        self.alphas_cumprod = inv.alphas_cumprod.clone()
        self.final_alpha_cumprod = self.alphas_cumprod[0].clone()
        self.config = types.SimpleNamespace(**SCHED_CFG)
        self.timesteps, self.n = None, None

    def set_timesteps(self, n, device=None):
        self.n = n
        r = self.config.num_train_timesteps // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * r).round()[::-1].copy().astype(np.int64) + self.config.steps_offset)

    def _prev(self, t):
        p = int(t) - self.config.num_train_timesteps // self.n
        return self.alphas_cumprod[p].double() if p >= 0 else self.final_alpha_cumprod.double()

    def step(self, model_output, timestep, sample, **unused):
        a_t, a_p = self.alphas_cumprod[int(timestep)].double(), self._prev(timestep)
        e, x = model_output.double(), sample.double()
        x0 = (x - (1 - a_t).sqrt() * e) / a_t.sqrt()
        return {"prev_sample": (a_p.sqrt() * x0 + (1 - a_p).sqrt() * e).to(sample.dtype)}


class RefDDPM(RefDDIM):
    """diffusers-0.15.0 ``DDPMScheduler`` (variance "fixed_small", epsilon, no clipping): timesteps without ``steps_offset``;
    ``prev = c0 x0 + ct x + sqrt(var) n`` with n = ``randn(model_output.shape, dtype=model_output.dtype)`` from the global RNG."""

    def set_timesteps(self, n, device=None):
        self.n = n
        r = self.config.num_train_timesteps // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * r).round()[::-1].copy().astype(np.int64))

    def step(self, model_output, timestep, sample, generator=None, **unused):
        t = int(timestep)
        p = t - self.config.num_train_timesteps // self.n
        a_t = self.alphas_cumprod[t].double()
        a_p = self.alphas_cumprod[p].double() if p >= 0 else torch.tensor(1.0, dtype=torch.float64)
        alpha_t = a_t / a_p
        beta_t = 1 - alpha_t
        e, x = model_output.double(), sample.double()
        x0 = (x - (1 - a_t).sqrt() * e) / a_t.sqrt()
        prev = a_p.sqrt() * beta_t / (1 - a_t) * x0 + alpha_t.sqrt() * (1 - a_p) / (1 - a_t) * x
        if t > 0:
            noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            var = torch.clamp((1 - a_p) / (1 - a_t) * beta_t, min=1e-20)
            prev = prev + var.sqrt() * noise.double()
        return {"prev_sample": prev.to(sample.dtype)}


# ----------------------------------------------------------------------------------------------- components
class ToyTextEmbedder(nn.Module):
    """``TextEmbedder.forward(text_prompts=..., train=False)`` (``seine/models/clip.py:60-122``): tokenise to 77 tokens, last hidden
    state of the text model."""

    def __init__(self, dim):
        super().__init__()
        self.tok, self.enc = rp.ToyTokenizer(), rp.ToyTextEncoder(dim)

    def forward(self, text_prompts, train=False, force_drop_ids=None):
        ids = self.tok(text_prompts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        return self.enc(ids)[0]


class _Fp32VAE(nn.Module):
    def __init__(self):
        super().__init__()
        self.toy = rcp.ToyVAE()

    def encode(self, x):
        return self.toy.encode(x.float())

    def decode(self, z):
        return self.toy.decode(z.float())


class _Fp32UNet(nn.Module):
    """The reference UNet in fp32 behind the dtype surface of the fp16 module the runners drive: fp16 in, fp16 out."""

    def __init__(self, unet):
        super().__init__()
        self.inner = unet

    def __getattr__(self, k):            # ``model.unet.up_blocks`` ... for the hook functions
        try:
            return super().__getattr__(k)
        except AttributeError:
            return getattr(super().__getattr__("inner"), k)

    def forward(self, x, t, encoder_hidden_states=None):
        y = self.inner(x.float(), t, encoder_hidden_states=encoder_hidden_states.float()).sample.half()
        return _Out(sample=y)


class _Out(dict):
    @property
    def sample(self):
        return self["sample"]


# ----------------------------------------------------------------------------------------------- module loading
def load_reference_seine_runners():
    """(run_ddim_inversion module, run_pnp_edit module, unet module, pnp_utils module)."""
    import transformers  # noqa: F401
    from anyv2v_amd.config import OmegaConf   # (attribute-access config objects only)
    att, ublocks, res, pnp, Rotary = ref_stubs.load_reference_seine_decoder(with_unet=True)
    before = set(sys.modules)
    saved = {k: sys.modules.get(k) for k in ("datasets", "models", "diffusion", "pnp_utils", "seine_utils", "omegaconf")}
    ref_stubs.install_stubs()
    root = os.path.join(ref_stubs.REFERENCE_ROOT, "seine")
    m = ref_stubs._mod
    try:
        m("torchvision")
        m("torchvision.io", read_video=None, write_video=None)
        tf = m("torchvision.transforms", Compose=rcp._Compose, Normalize=rcp._Normalize, ToPILImage=None, RandomCrop=None, RandomResizedCrop=None)
        sys.modules["torchvision"].transforms = tf
        m("omegaconf", OmegaConf=OmegaConf)
        m("diffusers", AutoencoderKL=object, UNet2DConditionModel=object, DDIMScheduler=object, DDPMScheduler=object,
          StableDiffusionPipeline=object)
        m("diffusion", create_diffusion=None)
        sys.modules["pnp_utils"] = pnp
        mods = types.ModuleType("models")
        mods.__path__ = []
        sys.modules["models"] = mods
        sys.modules["models.unet"] = ublocks.unet
        m("models.clip", TextEmbedder=ToyTextEmbedder)

        def load(name, path):
            spec = importlib.util.spec_from_file_location(name, path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
            return mod
        ds = types.ModuleType("datasets")
        ds.__path__ = []
        sys.modules["datasets"] = ds
        ds.video_transforms = load("datasets.video_transforms", os.path.join(root, "datasets", "video_transforms.py"))
        load("seine_utils", os.path.join(root, "seine_utils.py"))
        inv = load("_ref_seine_run_ddim_inversion", os.path.join(root, "run_ddim_inversion.py"))
        ed = load("_ref_seine_run_pnp_edit", os.path.join(root, "run_pnp_edit.py"))
    finally:
        for k in set(sys.modules) - before:
            if k.split(".")[0] in ("torchvision", "diffusers", "datasets", "models", "diffusion", "pnp_utils", "seine_utils", "omegaconf"):
                del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
    return inv, ed, ublocks.unet, pnp


class _Logger:
    def __getattr__(self, k):
        return lambda *a, **kw: None


@torch.no_grad()
def run_reference_job(unet_cfg, fill_weights, weight_seed, frames, edited, cfg_inv, cfg_edit, work_dir):
    """Stage 1 and stage 2 of the reference on ONE clip, driven as ``run_ddim_inversion.py:276-330`` / ``run_pnp_edit.py:346-376`` drive
    the two classes.  ``cfg_inv`` / ``cfg_edit``: config objects with the keys of ``seine/configs/*.yaml`` (paths are filled in here).
    Returns the trajectory files, the reconstruction and the edited frames (uint8, [1, f, h, w, c]) and latents."""
    inv_mod, ed_mod, unet_mod, pnp = load_reference_seine_runners()
    dev = torch.device("cpu")
    dim = unet_cfg["cross_attention_dim"]
    work_dir = str(work_dir)
    clip_dir = os.path.join(work_dir, "clip")
    os.makedirs(clip_dir, exist_ok=True)
    for i, f in enumerate(frames):
        f.save(os.path.join(clip_dir, "%05d.png" % i))
    edited_path = os.path.join(work_dir, "edited.png")
    edited.save(edited_path)

    def make_unet():
        return _Fp32UNet(fill_weights(unet_mod.UNet3DConditionModel(**unet_cfg), weight_seed).eval())

    # ---- stage 1 (run_ddim_inversion.py main + __init__)
    cfg_inv.src_video_path, cfg_inv.output_dir = clip_dir, os.path.join(work_dir, "ddim-inversion", "default")
    for mod, cfg in ((inv_mod, cfg_inv), (ed_mod, cfg_edit)):
        mod.logger, mod.device, mod.config = _Logger(), dev, cfg
    pnp.seed_everything(cfg_inv.seed)
    toy = RefDDIM()
    toy.set_timesteps(cfg_inv.n_save_steps)
    timesteps_to_save, _ = inv_mod.get_timesteps(toy, num_inference_steps=cfg_inv.n_save_steps, strength=1.0)
    from pathlib import Path
    save_path = os.path.join(cfg_inv.output_dir, cfg_inv.model_name, Path(cfg_inv.src_video_path).stem, f"steps_{cfg_inv.n_steps}",
                             f"nframes_{cfg_inv.n_frame_to_invert}")
    os.makedirs(os.path.join(save_path, "ddim_latents"), exist_ok=True)
    inv_mod.add_dict_to_yaml_file(file_path=os.path.join(save_path, "inversion_prompts.yaml"), key=Path(cfg_inv.src_video_path).stem,
                                  value=cfg_inv.inversion_prompt)
    p1 = object.__new__(inv_mod.SEINEDDIMInversionPipeline)
    nn.Module.__init__(p1)
    p1.device, p1.unet, p1.vae, p1.text_encoder, p1.scheduler = dev, make_unet(), _Fp32VAE(), ToyTextEmbedder(dim), RefDDIM()
    p1.paths, p1.frames = pnp.load_video_frames(cfg_inv.src_video_path, cfg_inv.n_frame_to_invert)
    p1.transform_video = inv_mod.transforms.Compose([inv_mod.video_transforms.ToTensorVideo(),
                                                     inv_mod.video_transforms.ResizeVideo(tuple(cfg_inv.image_size)),
                                                     inv_mod.transforms.Normalize(mean=[0.5, 0.5, 0.5], std=[0.5, 0.5, 0.5], inplace=True)])
    p1.frames = p1.transform_video(p1.frames)
    lat0 = p1.vae.encode(p1.frames.to(torch.float16)).latent_dist.sample().mul_(0.18215)
    from einops import rearrange
    p1.latent_at_0 = rearrange(lat0, "(b f) c h w -> b c f h w", b=1).contiguous().to(torch.float16)   # (fp16 storage, as the fp16 VAE's output)
    lat0_keep = p1.latent_at_0.clone()
    cap1 = []
    orig1 = p1.decode_latents
    p1.decode_latents = lambda lat: (cap1.append(lat.detach().clone()), orig1(lat))[1]
    recon_frames = p1.extract_ddim_latents(cfg_inv, timesteps_to_save, save_path)
    lat_dir = os.path.join(save_path, "ddim_latents")
    files = {int(f.split("_")[-1].split(".")[0]): torch.load(os.path.join(lat_dir, f)) for f in sorted(os.listdir(lat_dir))}

    # ---- stage 2 (run_pnp_edit.py main + __init__)
    cfg_edit.src_video_path = clip_dir + ".mp4"        # (the runner strips the suffix again: Path(...).parent / stem)
    cfg_edit.edited_first_frame_path, cfg_edit.ddim_inversion_dir = edited_path, cfg_inv.output_dir
    # (the real constructor builds the UNet AFTER seeding and so consumes an init-dependent number of RNG draws before the first
    # DDPM noise; here the model exists before the seed is set, so that the noise stream is a function of the seed alone -- the
    # native side of the comparison does the same)
    unet2, vae2, text2 = make_unet(), _Fp32VAE(), ToyTextEmbedder(dim)
    pnp.seed_everything(cfg_edit.seed)
    p2 = object.__new__(ed_mod.SEINEPnPPipeline)
    nn.Module.__init__(p2)
    p2.config, p2.device, p2.unet = cfg_edit, dev, unet2
    p2.latent_h, p2.latent_w, p2.latent_c, p2.n_frames = cfg_edit.image_size[0] // 8, cfg_edit.image_size[1] // 8, 4, cfg_edit.n_frames
    p2.vae, p2.text_encoder = vae2, text2
    p2.scheduler = RefDDIM() if cfg_edit.sample_method == "ddim" else RefDDPM()
    p2.ddim_latents_path = p2.get_ddim_latents_path()
    p2.ddim_latents_at_T = pnp.load_ddim_latents_at_T(p2.ddim_latents_path).to(torch.float16).to(dev)
    p2.ddim_inversion_prompt = p2.get_ddim_inversion_prompt()
    p2.edited_1st_frame = torch.as_tensor(np.array(__import__("PIL.Image").Image.open(edited_path).convert("RGB"), dtype=np.uint8, copy=True)).unsqueeze(0)
    p2.src_video_paths, p2.src_video_frames = pnp.load_video_frames(os.path.join(Path(cfg_edit.src_video_path).parent, Path(cfg_edit.src_video_path).stem),
                                                                   cfg_edit.n_frame_inverted)
    p2.transform_video = ed_mod.transforms.Compose([ed_mod.video_transforms.ToTensorVideo(),
                                                    ed_mod.video_transforms.ResizeVideo(tuple(cfg_edit.image_size)),
                                                    ed_mod.transforms.Normalize(mean=[0.5, 0.5, 0.5], std=[0.5, 0.5, 0.5], inplace=True)])
    p2.src_video_frames = p2.transform_video(p2.src_video_frames)
    p2.scheduler.set_timesteps(cfg_edit.n_steps)
    if cfg_edit.enable_pnp:
        p2.scheduler.set_timesteps(cfg_edit.n_steps)
        p2.init_pnp()
    cap = []
    orig = p2.decode_latents
    p2.decode_latents = lambda lat: (cap.append(lat.detach().clone()), orig(lat))[1]
    edited_frames = p2.edit_video(cfg_edit)
    return dict(lat0=lat0_keep, files=files, save_path=save_path, recon_frames=recon_frames, recon_lat=cap1[-1], edited_frames=edited_frames, edit_lat=cap[-1],
                edit_ts=[int(t) for t in p2.scheduler.timesteps], inv_ts=sorted(files), clip_dir=clip_dir, edited_path=edited_path)
