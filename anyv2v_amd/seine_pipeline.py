"""The SEINE backend's two runner classes on the HIP kernels: ``SEINEDDIMInversionPipeline`` (``seine/run_ddim_inversion.py:59-273``) and
``SEINEPnPPipeline`` (``seine/run_pnp_edit.py:44-343``) with the reference's method names, config keys and on-disk layout
(``<output_dir>/seine/<clip>/steps_<n>/nframes_<f>/ddim_latents/ddim_latents_{t}.pt``, ``inversion_prompts.yaml``), around
``anyv2v_amd.seine.UNet3DConditionModel`` and the five hook functions of ``anyv2v_amd.seine``.

SEINE conditions on the first frame through the UNet INPUT: 9 channels = noisy latents | mask (0 on frame 0, 1 elsewhere) | VAE
latents of the clip with every frame but the first blanked (``:202-243``).  Both runners do their own DDIM arithmetic from the
scheduler's ``alphas_cumprod``; the edit samples with DDIM or -- the shipped default -- ancestral DDPM steps
(``configs/pnp_edit.yaml:27``), reading the source latents of timestep t (DDIM) or t + 1 (DDPM).  One step here = one UNet forward +
one ``anyv2v_guided_step[_noise]_f16`` launch (guidance, x0 / eps, step, noise).

Batch rows of a PnP step: [source | cond, uncond] -- conditional FIRST, unlike the other two families (``:179-180,195``).
"""
from __future__ import annotations

import glob
import logging
import os
from pathlib import Path

import numpy as np
import torch
import yaml
from PIL import Image

from . import ops
from . import seine as sn
from .schedulers import SEINE_SCHEDULER_CONFIG, DDIMScheduler, DDPMScheduler

logger = logging.getLogger(__name__)

# Stable Diffusion 1.4's ``unet/config.json`` as ``UNet3DConditionModel.from_pretrained_2d(..., use_concat=True)`` rewrites it
# (``seine/models/unet.py:560-606``: 3-D block names, 9 input channels).  The file is not in the reference tree.
SEINE_UNET_CONFIG = dict(sample_size=64, in_channels=9, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                         norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8, use_linear_projection=False)


# ------------------------------------------------------------------------------------------------- helpers (pnp_utils / seine_utils)
def load_video_frames(frames_path, n_frames):
    """``seine/pnp_utils.py:46-53``: uint8 [f, c, h, w]."""
    paths = [f"{frames_path}/%05d.png" % i for i in range(n_frames)]
    if not os.path.exists(paths[0]):
        paths = [f"{frames_path}/%05d.jpg" % i for i in range(n_frames)]
    frames = [torch.as_tensor(np.array(Image.open(p).convert("RGB"), dtype=np.uint8, copy=True)).unsqueeze(0) for p in paths]
    return paths, torch.cat(frames, dim=0).permute(0, 3, 1, 2)


def load_ddim_latents_at_t(t, ddim_latents_path):
    p = os.path.join(ddim_latents_path, f"ddim_latents_{int(t)}.pt")
    assert os.path.exists(p), f"Missing latents at t {t} path {p}"
    return torch.load(p, map_location="cpu")


def load_ddim_latents_at_T(ddim_latents_path):
    noisest = max(int(x.split("_")[-1].split(".")[0]) for x in glob.glob(os.path.join(ddim_latents_path, "ddim_latents_*.pt")))
    return torch.load(os.path.join(ddim_latents_path, f"ddim_latents_{noisest}.pt"), map_location="cpu")


def save_video_as_frames(video_path, img_size=(512, 512)):
    """``seine/pnp_utils.py:25-37``: the clip's frames, LANCZOS-resized to ``img_size`` (w, h), as ``<video dir>/<video name>/%05d.png``."""
    from .utils import convert_video_to_frames
    convert_video_to_frames(video_path, tuple(img_size), save_frames=True)


def load_imgs(data_path, n_frames, device="cuda", pil=False):
    """``seine/pnp_utils.py:90-103``: ``%05d.jpg`` (else ``.png``) frames as a float tensor [n, c, H, W] in [0, 1] (``ToTensor``)."""
    pils = []
    for i in range(n_frames):
        path = os.path.join(data_path, "%05d.jpg" % i)
        if not os.path.exists(path):
            path = os.path.join(data_path, "%05d.png" % i)
        pils.append(Image.open(path))
    imgs = torch.stack([torch.from_numpy(np.array(im, dtype=np.uint8, copy=True)).permute(2, 0, 1).float() / 255.0 for im in pils]).to(device)
    return (imgs, pils) if pil else imgs


def save_video(raw_frames, save_path, fps=10, scaling_255=False):
    """``seine/pnp_utils.py:106-118``: [f, c, h, w] (uint8 values, or [0, 1] with ``scaling_255``; truncated as ``.to(torch.uint8)`` does) ->
    an H.264 mp4 (the reference encodes with libx264 at crf 18; there is no encoder library here: raw I_PCM pictures, ``anyv2v_amd.mp4``)."""
    from .mp4 import write_mp4
    x = (raw_frames * 255) if scaling_255 else raw_frames
    x = x.to(torch.uint8).cpu().permute(0, 2, 3, 1).numpy()
    return write_mp4([Image.fromarray(fr) for fr in x], save_path, fps=fps)


def mask_generation_before(mask_type, shape, dtype, device):
    """``seine/seine_utils.py:5-29``: 0 = frame given, 1 = frame to generate."""
    b, f, c, h, w = shape
    if mask_type.startswith("first"):
        num = int(mask_type.split("first")[-1])
        m = torch.cat([torch.zeros(1, num, 1, 1, 1, dtype=dtype, device=device), torch.ones(1, f - num, 1, 1, 1, dtype=dtype, device=device)], 1)
        return m.expand(b, -1, c, h, w)
    if mask_type.startswith("all"):
        return torch.ones(b, f, c, h, w, dtype=dtype, device=device)
    if mask_type.startswith("onelast"):
        num = int(mask_type.split("onelast")[-1])
        z, o = torch.zeros(1, 1, 1, 1, 1, dtype=dtype, device=device), torch.ones(1, f - 2 * num, 1, 1, 1, dtype=dtype, device=device)
        return torch.cat([z] * num + [o] + [z] * num, dim=1).expand(b, -1, c, h, w)
    raise ValueError(f"Invalid mask type: {mask_type}")


def transform_video(frames_u8: torch.Tensor, image_size):
    """``ToTensorVideo -> ResizeVideo(image_size) -> Normalize(0.5, 0.5)`` (``seine/datasets/video_transforms.py:146-159,48-51``,
    ``run_ddim_inversion.py:104-110``): uint8 [f, c, h, w] -> [-1, 1], bilinear ``align_corners=False`` to (height, width)."""
    if frames_u8.dtype != torch.uint8:
        raise TypeError("clip tensor should have data type uint8. Got %s" % str(frames_u8.dtype))
    x = frames_u8.float() / 255.0
    x = torch.nn.functional.interpolate(x, size=tuple(int(s) for s in image_size), mode="bilinear", align_corners=False)
    return (x - 0.5) / 0.5


def build_components(config, device, random_init_seed=None):
    """(unet, vae, text_encoder, scheduler config): a local Stable-Diffusion-1.4 directory + ``seine.pt`` when they exist
    (``run_ddim_inversion.py:65-88``), else -- there is no network -- random weights of the configured architecture
    (``random_init_seed`` / ANYV2V_RANDOM_INIT_SEED) with the synthetic VAE / text encoder."""
    import json
    if not config.get("use_fp16", True):
        # (``run_ddim_inversion.py:83-85`` keeps the model in fp32 then; ``enable_xformers_memory_efficient_attention`` only picks an
        # attention implementation there and is accepted silently)
        logger.warning("use_fp16: False -- the HIP kernels compute in fp16 with fp32 accumulation whatever this says")
    sd_path, ckpt = str(config.get("sd_path", "")), str(config.get("ckpt_path", ""))
    cfg = dict(SEINE_UNET_CONFIG)
    cj = os.path.join(sd_path, "unet", "config.json")
    if os.path.isfile(cj):
        with open(cj) as f:
            raw = json.load(f)
        cfg.update({k: raw[k] for k in ("block_out_channels", "layers_per_block", "norm_num_groups", "norm_eps", "cross_attention_dim",
                                        "attention_head_dim", "use_linear_projection", "sample_size") if k in raw})
    unet = sn.UNet3DConditionModel(**cfg)
    if os.path.isfile(ckpt):
        state = torch.load(ckpt, map_location="cpu")
        unet.load_state_dict(state["ema"] if "ema" in state else state, strict=True)
    else:
        seed = random_init_seed
        if seed is None and os.environ.get("ANYV2V_RANDOM_INIT_SEED") is not None:
            seed = int(os.environ["ANYV2V_RANDOM_INIT_SEED"])
        if seed is None:
            raise FileNotFoundError(f"no SEINE checkpoint at {ckpt!r} and no network; pass random_init_seed= (or ANYV2V_RANDOM_INIT_SEED)")
        from .consisti2v_pipeline import init_random_weights_
        init_random_weights_(unet, seed)
    unet.to(device)
    from .encoders import NativeVAE, SyntheticTextEncoder, SyntheticVAE
    vpath = os.path.join(sd_path, "vae", "diffusion_pytorch_model.safetensors")
    if os.path.isfile(vpath):
        from safetensors.torch import load_file
        vae = NativeVAE(state_dict=load_file(vpath)).to(device)
    else:
        vae = SyntheticVAE()
    text = None
    if os.path.isdir(os.path.join(sd_path, "text_encoder")):
        import types
        from .encoders import attach_native_text_encoder
        holder = types.SimpleNamespace()
        if not attach_native_text_encoder(holder, sd_path):
            raise FileNotFoundError(f"{sd_path}/text_encoder exists but {sd_path}/tokenizer does not: cannot build the text encoder")
        text = holder.text_encoder.to(device)
    if text is None:
        if os.path.isfile(ckpt):
            logger.warning(f"{ckpt}: UNet weights found but no text encoder under {sd_path!r} -- using the synthetic stand-in")
        text = SyntheticTextEncoder(dim=int(cfg["cross_attention_dim"]))
    return unet, vae, text


def _scheduler(config, cls):
    kw = dict(SEINE_SCHEDULER_CONFIG)
    kw.update(beta_start=config.beta_start, beta_end=config.beta_end, beta_schedule=config.beta_schedule)
    return cls(**kw)


class _Base:
    def _init_components(self, device, config, unet, vae, text_encoder, random_init_seed):
        self.device, self.config = torch.device(device), config
        if unet is None:
            unet, vae, text_encoder = build_components(config, self.device, random_init_seed)
        self.unet, self.vae, self.text_encoder = unet, vae, text_encoder

    def _embed(self, prompts):
        return self.text_encoder.encode(prompts, self.device, None).to(torch.float16)

    def _encode(self, pixels):
        """[n, 3, H, W] in [-1, 1] -> VAE latents x 0.18215, one posterior sample per frame batch."""
        return self.vae.encode_pixels(pixels.to(torch.float16), self.device)

    @torch.no_grad()
    def decode_latents(self, latents):
        """``:113-121``: uint8 [b, f, h, w, c] on the host (``(x / 2 + 0.5) * 255 + 0.5`` clamped, truncated)."""
        video = self.vae.decode_video(latents.to(torch.float16), decode_chunk_size=None)          # [1, 3, F, H, W] in [-1, 1]
        video = video.permute(0, 2, 3, 4, 1)
        return ((video / 2 + 0.5) * 255).add_(0.5).clamp_(0, 255).to(dtype=torch.uint8).cpu().contiguous()

    def _first_frame_condition(self, first_frame_pixels, n_frames, latent_hw):
        """``extract_ddim_latents`` / ``compute_masked_video_latents_at_0``: the clip with all frames but the first set to ZERO pixels,
        encoded frame by frame; the mask at latent resolution."""
        dev = self.device
        video = torch.cat([first_frame_pixels] + [torch.zeros_like(first_frame_pixels)] * (n_frames - 1), dim=0).to(dev).unsqueeze(0)
        return self._condition_from_video(video, latent_hw)

    def _condition_from_video(self, video, latent_hw):
        """[1, f, c, H, W] pixels -> (mask [1, 1, f, h, w], latents of the clip with every frame but the first zeroed [1, 4, f, h, w])."""
        dev = self.device
        if video.shape[0] != 1:
            raise NotImplementedError(f"one clip per call: video batch {video.shape[0]}")
        video = video.to(dev)
        mask = mask_generation_before("first1", video.shape, video.dtype, dev)                       # b f c h w
        masked = (video * (mask == 0)).to(torch.float16)
        mask = mask.to(torch.float16)
        lat = self._encode(masked[0])                                                                 # [f, 4, h, w]
        lat = lat.permute(1, 0, 2, 3)[None].contiguous()
        mask = torch.nn.functional.interpolate(mask[:, :, 0, :], size=latent_hw).unsqueeze(1)        # [1, 1, f, h, w]
        return mask.contiguous(), lat

    def _unet_input(self, rows, mask, masked_rows):
        return torch.cat([torch.cat([x.to(torch.float16), mask, mv], dim=1) for x, mv in zip(rows, masked_rows)]).contiguous()


class SEINEDDIMInversionPipeline(_Base):
    def __init__(self, device, config, unet=None, vae=None, text_encoder=None, random_init_seed=None):
        self._init_components(device, config, unet, vae, text_encoder, random_init_seed)
        self.scheduler = _scheduler(config, DDIMScheduler)
        self.paths, frames = load_video_frames(config.src_video_path, config.n_frame_to_invert)
        self.frames = transform_video(frames, config.image_size)
        lat = self._encode(self.frames)
        self.latent_at_0 = lat.permute(1, 0, 2, 3)[None].contiguous()

    def _coef(self, t, other):
        ac = self.scheduler.alphas_cumprod
        a_t = float(ac[int(t)])
        a_o = float(ac[int(other)]) if other is not None else float(self.scheduler.final_alpha_cumprod)
        return a_t ** 0.5, (1 - a_t) ** 0.5, a_o ** 0.5, (1 - a_o) ** 0.5

    @torch.no_grad()
    def ddim_inversion(self, cond, latent_frames, masked_video, mask, save_path, batch_size, save_latents=True, timesteps_to_save=None):
        """``:127-168``: x at the level below t -> level t, eps evaluated at (x, t); files for the timesteps in ``timesteps_to_save``."""
        timesteps = [int(t) for t in reversed(self.scheduler.timesteps)]
        keep = set(int(t) for t in (timesteps_to_save if timesteps_to_save is not None else timesteps))
        x = latent_frames.to(torch.float16).contiguous()
        ehs = cond.to(torch.float16).contiguous()
        for i, t in enumerate(timesteps):
            mu, sigma, mu_prev, sigma_prev = self._coef(t, timesteps[i - 1] if i > 0 else None)
            eps = self.unet(self._unet_input([x], mask, [masked_video]), t, encoder_hidden_states=ehs).sample.contiguous()
            x = ops.guided_step(eps, x, (mu_prev, sigma_prev, mu, sigma), b_txt=0, prediction=ops.PRED_EPSILON)
            if save_latents and t in keep:
                torch.save(x.cpu(), os.path.join(save_path, "ddim_latents", f"ddim_latents_{t}.pt"))
                logger.info(f"[INFO] saved noisy latents at t={t} to {save_path}/ddim_latents/ddim_latents_{t}.pt")
        return x

    @torch.no_grad()
    def ddim_sample(self, x, cond, masked_video, mask, batch_size):
        """``:171-199``."""
        timesteps = [int(t) for t in self.scheduler.timesteps]
        x = x.to(torch.float16).contiguous()
        ehs = cond.to(torch.float16).contiguous()
        for i, t in enumerate(timesteps):
            mu, sigma, mu_prev, sigma_prev = self._coef(t, timesteps[i + 1] if i < len(timesteps) - 1 else None)
            eps = self.unet(self._unet_input([x], mask, [masked_video]), t, encoder_hidden_states=ehs).sample.contiguous()
            x = ops.guided_step(eps, x, (mu, sigma, mu_prev, sigma_prev), b_txt=0, prediction=ops.PRED_EPSILON)
        return x

    @torch.no_grad()
    def extract_ddim_latents(self, config, timesteps_to_save, save_path):
        """``:202-273``: inversion (files), then the DDIM reconstruction from the last latents; uint8 frames [1, f, h, w, c]."""
        latent_hw = (config.image_size[0] // 8, config.image_size[1] // 8)
        mask, masked_video = self._first_frame_condition(self.frames[0].unsqueeze(0), len(self.frames), latent_hw)
        self.scheduler.set_timesteps(config.n_steps)
        text_embeddings = self._embed(config.inversion_prompt)
        os.makedirs(os.path.join(save_path, "ddim_latents"), exist_ok=True)
        x_T = self.ddim_inversion(cond=text_embeddings, latent_frames=self.latent_at_0, masked_video=masked_video, mask=mask, save_path=save_path,
                                  batch_size=config.batch_size, save_latents=True, timesteps_to_save=timesteps_to_save)
        x_0 = self.ddim_sample(x=x_T, cond=text_embeddings, masked_video=masked_video, mask=mask, batch_size=config.batch_size)
        self.reconstructed_latents = x_0
        return self.decode_latents(x_0)


class SEINEPnPPipeline(_Base):
    def __init__(self, device, config, unet=None, vae=None, text_encoder=None, random_init_seed=None):
        self._init_components(device, config, unet, vae, text_encoder, random_init_seed)
        self.latent_h, self.latent_w, self.latent_c = config.image_size[0] // 8, config.image_size[1] // 8, 4
        self.n_frames = config.n_frames
        if config.sample_method == "ddim":
            self.scheduler = _scheduler(config, DDIMScheduler)
        elif config.sample_method == "ddpm":
            self.scheduler = _scheduler(config, DDPMScheduler)
        else:
            raise NotImplementedError(config.sample_method)
        self.ddim_latents_path = self.get_ddim_latents_path()
        self.ddim_latents_at_T = load_ddim_latents_at_T(self.ddim_latents_path).to(torch.float16).to(self.device)
        self.ddim_inversion_prompt = self.get_ddim_inversion_prompt()
        logger.info(f"ddim_inversion_prompt: {self.ddim_inversion_prompt}")
        self.edited_1st_frame = torch.as_tensor(np.array(Image.open(config.edited_first_frame_path).convert("RGB"), dtype=np.uint8, copy=True)).unsqueeze(0)
        frames_dir = os.path.join(Path(config.src_video_path).parent, Path(config.src_video_path).stem)
        self.src_video_paths, frames = load_video_frames(frames_dir, config.n_frame_inverted)
        self.src_video_frames = transform_video(frames, config.image_size)

    def get_ddim_inversion_prompt(self):
        """``:134-138``."""
        with open(os.path.join(str(Path(self.ddim_latents_path).parent), "inversion_prompts.yaml"), "r") as f:
            return yaml.safe_load(f)[f"{Path(self.config.src_video_path).stem}"]

    def get_ddim_latents_path(self):
        """``:140-157``: the inversion of this clip with the most frames; ``n_frames`` is cut to it (and to a multiple of batch_size)."""
        config = self.config
        base = os.path.join(config.ddim_inversion_dir, config.model_name, Path(config.src_video_path).stem, f"steps_{config.n_ddim_inversion_steps}")
        cands = [x for x in glob.glob(f"{base}/*") if "." not in Path(x).name]
        n_frames = [int([p for p in c.split("/") if "nframes" in p][0].split("_")[1]) for c in cands]
        best = cands[int(np.argmax(n_frames))]
        config.n_frames = min(max(n_frames), config.n_frames)
        if config.n_frames % config.batch_size != 0:
            config.n_frames = config.n_frames - (config.n_frames % config.batch_size)
        return os.path.join(best, "ddim_latents")

    def init_pnp(self, override_dict=None):
        """``:209-243``."""
        c = self.config
        if override_dict is not None:
            c.pnp_f_t, c.pnp_spatial_attn_t = override_dict["conv_inject"], override_dict["attn_inject"]
            c.pnp_cross_attn_t, c.pnp_temp_attn_t = override_dict["cross_inject"], override_dict["temp_inject"]
        pick = lambda ratio: (lambda n: self.scheduler.timesteps[:n] if n >= 0 else [])(int(c.n_steps * ratio))
        self.conv_injection_timesteps = pick(c.pnp_f_t)
        self.spatial_attn_qk_injection_timesteps = pick(c.pnp_spatial_attn_t)
        self.cross_attn_qk_injection_timesteps = pick(c.pnp_cross_attn_t)
        self.temp_attn_qk_injection_timesteps = pick(c.pnp_temp_attn_t)
        sn.register_conv_injection(self, self.conv_injection_timesteps)
        sn.register_spatial_attention_pnp(self, self.spatial_attn_qk_injection_timesteps)
        sn.register_cross_attention_pnp(self, self.cross_attn_qk_injection_timesteps)
        sn.register_temp_attention_pnp(self, self.temp_attn_qk_injection_timesteps)

    def compute_masked_video_latents_at_0(self, config, video_input):
        """``:256-277``.  ``video_input``: the reference's [b, f, c, H, W] clip (only its first frame survives the mask), or just that
        first frame [1, c, H, W] (the clip is then completed with zero frames here instead of by the caller)."""
        if video_input.dim() == 5:
            return self._condition_from_video(video_input, (self.latent_h, self.latent_w))
        return self._first_frame_condition(video_input, len(self.src_video_frames), (self.latent_h, self.latent_w))

    @torch.no_grad()
    def denoise_step(self, x, mask, masked_video, masked_src_video, t):
        """``:162-207``."""
        c = self.config
        t = int(t)
        ddpm = c.sample_method == "ddpm"
        if c.enable_pnp:
            src = load_ddim_latents_at_t(t + 1 if ddpm else t, self.ddim_latents_path).to(self.device)
            model_input = self._unet_input([src, x, x], mask, [masked_src_video, masked_video, masked_video])
            ehs = torch.cat([self.ddim_inversion_embeds, self.cond_embeds, self.uncond_embeds], dim=0)
            sn.register_time(self, t)
            b_cond, b_unc = 1, 2
        else:
            model_input = self._unet_input([x, x], mask, [masked_video, masked_video])
            ehs = torch.cat([self.cond_embeds, self.uncond_embeds], dim=0)
            b_cond, b_unc = 0, 1
        e = self.unet(model_input, t, encoder_hidden_states=ehs.contiguous()).sample.contiguous()
        guided = c.cfg_scale > 1.0
        kw = dict(b_txt=b_cond, b_unc=b_unc if guided else -1, g_txt=float(c.cfg_scale), prediction=ops.PRED_EPSILON)
        x = x.to(torch.float16).contiguous()
        if ddpm:
            sa_t, sb_t, cx, ce, sigma = self.scheduler.ancestral_coefficients(t)
            noise = self.scheduler.draw_noise(x, t)
            return ops.guided_step(e, x, (sa_t, sb_t, cx, ce), noise=noise, sigma=sigma, **kw)
        return ops.guided_step(e, x, self.scheduler.coefficients(t), **kw)

    def sample_loop(self, x, mask, masked_video, masked_src_video):
        """``:345-360``."""
        bs = self.config.batch_size
        for t in self.scheduler.timesteps:
            x = torch.cat([self.denoise_step(x[b:b + bs], mask[b:b + bs], masked_video[b:b + bs], masked_src_video[b:b + bs], t)
                           for b in range(0, len(x), bs)])
        return x

    @torch.no_grad()
    def edit_video(self, config):
        """``:265-343``."""
        first = transform_video(self.edited_1st_frame.permute(0, 3, 1, 2), config.image_size)
        mask, masked_edited = self.compute_masked_video_latents_at_0(config, first)
        _, masked_src = self.compute_masked_video_latents_at_0(config, self.src_video_frames[0].unsqueeze(0))
        if config.init_with_ddim_inversion:
            x_T = self.ddim_latents_at_T.to(self.device)
        else:
            x_T = torch.randn(1, self.latent_c, config.n_frames, self.latent_h, self.latent_w, dtype=torch.float16).to(self.device)
        if config.enable_pnp:
            emb = self._embed([self.ddim_inversion_prompt, config.prompt, config.negative_prompt])
            self.ddim_inversion_embeds, self.cond_embeds, self.uncond_embeds = emb.chunk(3, dim=0)
        else:
            self.cond_embeds, self.uncond_embeds = self._embed([config.prompt, config.negative_prompt]).chunk(2, dim=0)
        try:
            x_0 = self.sample_loop(x_T.to(torch.float16), mask, masked_edited, masked_src)
        finally:
            if config.enable_pnp:
                sn.clear_time(self)
        self.edited_latents = x_0
        return self.decode_latents(x_0)
