"""``ConditionalVideoEditingPipeline`` of the ConsistI2V backend (``consisti2v/consisti2v/pipelines/pipeline_video_editing.py:128-1579``)
on the HIP kernels: ``encode_vae_video`` (``:1226-1258``), ``invert`` (``:715-968``), ``__call__`` (DDIM reconstruction / sampling from
a stored latent, ``:469-711``) and ``sample_with_pnp`` (``:1261-1576``) with the reference's argument names, defaults and file
formats, around ``anyv2v_amd.consisti2v.VideoLDMUNet3DConditionModel`` and the hook functions of ``anyv2v_amd.consisti2v``.

What one denoising step is here: one UNet forward over all branches ([source | negative, editing] for PnP), then ONE elementwise
kernel for guidance + scheduler step (``anyv2v_guided_step_f16``).  Pre-processing mirrors the reference's torchvision chains on
torch tensors: ``ToTensor -> Resize(height) [shorter edge, bilinear, no antialias] -> CenterCrop -> Normalize(0.5, 0.5)`` for
``encode_vae_video`` / ``invert`` (``:788-793,1233-1238``) and ``ToTensor -> Resize((height, width)) -> Normalize`` for ``__call__`` /
``sample_with_pnp`` (``:541-547,1351-1357``).

Kept from the reference because a drop-in must produce the same numbers:

* ``sample_with_pnp`` builds the SOURCE branch's first-frame latent from the EDITED first frame (``:1426`` appends ``first_frame``,
  the last tensor of the loop above it, not the image it just opened), encoded a second time (a second posterior sample);
* frame 0 of the start latents is the "noisy first frame" of the image-unconditional branch and otherwise dropped (``:1478-1480``):
  the UNet denoises ``video_length - 1`` frames and sees the clean first-frame latent as frame 0.

``use_frameinit`` (FFT noise re-initialisation, ``frameinit_utils.py``) is built (``init_filter``, host-side FFT once per clip).
``camera_motion`` (pan / zoom pseudo clips for FrameInit) is built as well, and so are ``guidance_rescale`` (text guidance) and
``eta`` (forward DDIM scheduler).  Not built (the AnyV2V runners never need them): several clips per call
(``num_videos_per_prompt``, prompt lists), PnP with image guidance or without text guidance
(the reference's hooks split the batch in three: ``consisti2v/pnp_utils.py:96,188,296``).
"""
from __future__ import annotations

import logging
import math
import os
from typing import List, Optional, Union

import numpy as np
import torch
from PIL import Image

from . import consisti2v as c2
from . import ops
from .schedulers import CONSISTI2V_SCHEDULER_CONFIG, DDIMInverseScheduler, DDIMScheduler
from .utils import LatentTrajectory, capture_hip_graph, load_ddim_latents_at_t

logger = logging.getLogger(__name__)

# The released ``TIGER-Lab/ConsistI2V`` ``unet/config.json`` is not in the reference tree (``run_pnp_edit.py:52-55`` downloads it).
# This is the Stable-Diffusion-2.1-base layout ConsistI2V extends, with the video options its inference code reads
# (``videoldm_unet.py:96-147``); a local ``unet/config.json`` overrides it.
CONSISTI2V_UNET_CONFIG = dict(
    sample_size=32, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, norm_num_groups=32,
    cross_attention_dim=1024, attention_head_dim=(5, 10, 20, 20), use_linear_projection=True, use_temporal=True, n_frames=16,
    n_temp_heads=8, first_frame_condition_mode="concat", augment_temporal_attention=True, temp_pos_embedding="rotary",
    use_frame_stride_condition=True)


class AnimationPipelineOutput:
    def __init__(self, videos):
        self.videos = videos


# ------------------------------------------------------------------------------------------------- FrameInit (frameinit_utils.py)
def get_freq_filter(shape, device, filter_type, n, d_s, d_t):
    """``consisti2v/consisti2v/utils/frameinit_utils.py:36-141``: low-pass mask over the (T, H, W) frequency volume, centred
    (``fftshift`` order).  ``d^2 = ((d_s / d_t)(2t / T - 1))^2 + (2h / H - 1)^2 + (2w / W - 1)^2``; "gaussian": exp(-d^2 / (2 d_s^2)),
    "butterworth": 1 / (1 + (d^2 / d_s^2)^n), "ideal": d^2 <= 2 d_s (the reference compares with ``d_s*2``, kept), "box": a centred box
    of half-widths round(T // 2 * d_t), round(H // 2 * d_s) (the reference's function builds this mask and forgets to return it)."""
    T, H, W = shape[-3], shape[-2], shape[-1]
    mask = torch.zeros(shape)
    if d_s == 0 or d_t == 0:
        return mask.to(device)
    if filter_type == "box":
        ts, tt = round(int(H // 2) * d_s), round(T // 2 * d_t)
        cf, cr, cc = T // 2, H // 2, W // 2
        mask[..., cf - tt:cf + tt, cr - ts:cr + ts, cc - ts:cc + ts] = 1.0
        return mask.to(device)
    t = ((d_s / d_t) * (2 * torch.arange(T, dtype=torch.float64) / T - 1)) ** 2
    h = (2 * torch.arange(H, dtype=torch.float64) / H - 1) ** 2
    w = (2 * torch.arange(W, dtype=torch.float64) / W - 1) ** 2
    d2 = t[:, None, None] + h[None, :, None] + w[None, None, :]
    if filter_type == "gaussian":
        vol = torch.exp(-1 / (2 * d_s ** 2) * d2)
    elif filter_type == "butterworth":
        vol = 1 / (1 + (d2 / d_s ** 2) ** n)
    elif filter_type == "ideal":
        vol = (d2 <= d_s * 2).double()
    else:
        raise NotImplementedError(filter_type)
    return (mask + vol.float()).to(device)


def freq_mix_3d(x, noise, LPF):
    """``frameinit_utils.py:7-33``: low frequencies of the diffused first-frame video, high frequencies of the noise.  Once per clip on a
    [1, 4, F, h, w] tensor: computed on the host."""
    dev = x.device
    x, noise, LPF = x.float().cpu(), noise.float().cpu(), LPF.float().cpu()
    dims = (-3, -2, -1)
    xf = torch.fft.fftshift(torch.fft.fftn(x, dim=dims), dim=dims)
    nf = torch.fft.fftshift(torch.fft.fftn(noise, dim=dims), dim=dims)
    mixed = xf * LPF + nf * (1 - LPF)
    return torch.fft.ifftn(torch.fft.ifftshift(mixed, dim=dims), dim=dims).real.to(dev)


# ------------------------------------------------------------------------------------------------- pre-processing
def _to_tensor(img: Image.Image) -> torch.Tensor:
    return torch.from_numpy(np.asarray(img.convert("RGB"), dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0


def _resize(x: torch.Tensor, size) -> torch.Tensor:
    """torchvision ``Resize(size, antialias=None)`` of a tensor image: bilinear, ``align_corners=False``, no antialiasing; an int
    matches the shorter edge and scales the longer one to ``int(size * long / short)``."""
    H, W = x.shape[-2:]
    if isinstance(size, int):
        short, long = (H, W) if H <= W else (W, H)
        new_long = int(size * long / short)
        size = (size, new_long) if H <= W else (new_long, size)
    size = tuple(int(s) for s in size)
    if size == (H, W):
        return x
    return torch.nn.functional.interpolate(x[None], size=size, mode="bilinear", align_corners=False, antialias=False)[0]


def _center_crop(x: torch.Tensor, h: int, w: int) -> torch.Tensor:
    H, W = x.shape[-2:]
    if H < h or W < w:
        raise ValueError(f"frame of {H} x {W} is smaller than the {h} x {w} crop")
    top, left = int(round((H - h) / 2.0)), int(round((W - w) / 2.0))
    return x[..., top:top + h, left:left + w]


def camera_motion_frames(x: torch.Tensor, motion: str, num_frames: int, crop_width: int, ratio: float = 1.5) -> torch.Tensor:
    """``pan_right`` / ``pan_left`` / ``zoom_in`` / ``zoom_out`` (``pipeline_video_editing.py:63-121``): a pseudo clip [f, 3, h, w] cut out
    of one pre-processed frame [3, H, W] -- a crop window that slides (pan) or shrinks / grows around the centre and is resized back
    (zoom; bilinear, no antialiasing).  ``zoom_out`` hands float crop sizes to ``torchvision``'s crop in the reference (floor division
    by the float ``ratio``), which only runs where slices accept them; the sizes are truncated to int here."""
    H, W = x.shape[-2:]
    out = []
    for i in range(num_frames):
        if motion == "pan_right":
            sx = int((W - crop_width) * (i / num_frames))
            out.append(x[..., :, sx:sx + crop_width])
        elif motion == "pan_left":
            sx = int((W - crop_width) * (1 - (i / num_frames)))
            out.append(x[..., :, sx:sx + crop_width])
        elif motion in ("zoom_in", "zoom_out"):
            m = min(W, H)
            if motion == "zoom_in":
                cs = m - int((m - m // ratio) * (i / num_frames))
            else:
                cs = int(m // ratio + int((m - m // ratio) * (i / num_frames)))
            sx, sy = int((W - cs) // 2), int((H - cs) // 2)
            out.append(_resize(x[..., sy:sy + cs, sx:sx + cs], (crop_width, crop_width)))
        else:
            raise NotImplementedError(f"camera_motion: {motion} is not implemented.")
    return torch.stack(out)


def pan_right(image, num_frames=16, crop_width=256):
    """``pipeline_video_editing.py:63-73`` (and its three neighbours): the module-level names of the camera-motion helpers."""
    return camera_motion_frames(image, "pan_right", num_frames, crop_width)


def pan_left(image, num_frames=16, crop_width=256):
    return camera_motion_frames(image, "pan_left", num_frames, crop_width)


def zoom_in(image, num_frames=16, crop_width=256, ratio=1.5):
    return camera_motion_frames(image, "zoom_in", num_frames, crop_width, ratio)


def zoom_out(image, num_frames=16, crop_width=256, ratio=1.5):
    return camera_motion_frames(image, "zoom_out", num_frames, crop_width, ratio)


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    """``pipeline_video_editing.py:50-61`` on torch tensors (the loop applies the same formula to its fp32 copies, ``_denoise``): the
    guided prediction rescaled to the standard deviation of the text branch, mixed back by ``guidance_rescale``."""
    dims = list(range(1, noise_pred_text.ndim))
    std_text = noise_pred_text.std(dim=dims, keepdim=True)
    std_cfg = noise_cfg.std(dim=dims, keepdim=True)
    return guidance_rescale * (noise_cfg * (std_text / std_cfg)) + (1 - guidance_rescale) * noise_cfg


def frame_to_pixels(img: Image.Image, height: int, width: int, crop: bool) -> torch.Tensor:
    """[1, 3, height, width] in [-1, 1].  ``crop``: the ``Resize(height) + CenterCrop`` chain; else ``Resize((height, width))``."""
    x = _to_tensor(img)
    x = _center_crop(_resize(x, height), height, width) if crop else _resize(x, (height, width))
    return ((x - 0.5) / 0.5)[None]


class _StepGraphs:
    """Static inputs of the UNet forward of one loop geometry + one captured HIP graph per injection state."""

    def __init__(self, unet, nb, latents, ehs, frame_stride):
        dev = latents.device
        self.unet = unet
        # OFF by default: measured at the released width (tools/consisti2v_bench.py, 10 steps, graphs kept across calls) replaying the
        # ~1400-node graph is no faster than the eager launches at 16 f x 256^2 (15.7 / 30.5 ms per inversion / PnP step against
        # 16.0 / 25.9) and slower at 512^2 (44.7 / 81.0 against 34.6 / 82.0) -- the forward of this family is many 5-20 us launches
        # and the per-node cost of a replay is of that order.  ANYV2V_CONSISTI2V_GRAPHS=1 switches it on.
        self.use_graphs = dev.type == "cuda" and os.environ.get("ANYV2V_CONSISTI2V_GRAPHS", "0") == "1"
        self.x = torch.empty((nb,) + tuple(latents.shape[1:]), dtype=torch.float16, device=dev)
        self.ehs = torch.empty((nb,) + tuple(ehs.shape[1:]), dtype=torch.float16, device=dev)
        self.ff = torch.empty((nb, latents.shape[1], 1) + tuple(latents.shape[3:]), dtype=torch.float16, device=dev)
        self.t_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        self.fs = None if frame_stride is None else torch.full((1,), float(frame_stride), dtype=torch.float32, device=dev)
        self.graphs, self.outs = {}, {}

    def _forward(self):
        return self.unet(self.x, self.t_buf, encoder_hidden_states=self.ehs, first_frame_latents=self.ff, frame_stride=self.fs).sample.contiguous()

    def run(self, key):
        if not self.use_graphs:
            return self._forward()
        if key not in self.graphs:
            self._forward()                               # warm-up outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with capture_hip_graph(g):
                self.outs[key] = self._forward()
            self.graphs[key] = g
        self.graphs[key].replay()
        return self.outs[key]


# ------------------------------------------------------------------------------------------------- the pipeline
class ConditionalVideoEditingPipeline:
    def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet: Optional[c2.VideoLDMUNet3DConditionModel] = None, scheduler=None):
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
        self.vae_scale_factor = 8
        self._device = torch.device("cpu")
        self.freq_filter = None
        self._engines = {}

    # ------------------------------------------------------------------ construction / plumbing
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=torch.float16, unet_config: Optional[dict] = None,
                        random_init_seed: Optional[int] = None, **kw):
        """``<path>/unet/config.json`` + ``diffusion_pytorch_model.safetensors`` (the reference's key naming), ``<path>/vae``,
        ``<path>/text_encoder`` + ``tokenizer`` and ``<path>/scheduler/scheduler_config.json`` when a local copy exists.  There is no
        network here: for the hub id without a local copy, random weights of the configured architecture are used when
        ``random_init_seed`` (or ANYV2V_RANDOM_INIT_SEED) is given, with the synthetic VAE / text encoder."""
        import json
        if torch_dtype != torch.float16:
            raise ValueError("the HIP kernels compute in fp16 (fp32 accumulate); torch_dtype must be torch.float16")
        root = str(pretrained_model_name_or_path)
        cfg_json = os.path.join(root, "unet", "config.json")
        cfg = dict(CONSISTI2V_UNET_CONFIG)
        if unet_config is not None:
            cfg = dict(unet_config)
        elif os.path.isfile(cfg_json):
            with open(cfg_json) as f:
                cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        unet = c2.VideoLDMUNet3DConditionModel(**cfg)
        wpath = os.path.join(root, "unet", "diffusion_pytorch_model.safetensors")
        if os.path.isfile(wpath):
            from safetensors.torch import load_file
            unet.load_state_dict(load_file(wpath), strict=True)
        else:
            seed = random_init_seed
            if seed is None and os.environ.get("ANYV2V_RANDOM_INIT_SEED") is not None:
                seed = int(os.environ["ANYV2V_RANDOM_INIT_SEED"])
            if seed is None:
                raise FileNotFoundError(f"no UNet weights under {root!r} (expected unet/diffusion_pytorch_model.safetensors) and no network; "
                                        "pass random_init_seed= (or ANYV2V_RANDOM_INIT_SEED) to run with random weights")
            init_random_weights_(unet, seed)
        pipe = cls(unet=unet, scheduler=DDIMScheduler.from_pretrained(root, subfolder="scheduler", **_sched_defaults(root)))
        vpath = os.path.join(root, "vae", "diffusion_pytorch_model.safetensors")
        if os.path.isfile(vpath):
            from safetensors.torch import load_file

            from .encoders import NativeVAE
            pipe.vae = NativeVAE(state_dict=load_file(vpath))
        if os.path.isdir(os.path.join(root, "text_encoder")):
            from .encoders import attach_native_text_encoder
            if not attach_native_text_encoder(pipe, root):
                raise FileNotFoundError(f"{root}/text_encoder exists but {root}/tokenizer does not: cannot build the text encoder")
        if pipe.vae is None or pipe.text_encoder is None:
            if os.path.isfile(wpath):
                logger.warning(f"{root}: UNet weights found but no {'vae' if pipe.vae is None else 'text_encoder'} -- using the synthetic stand-in")
            from .encoders import SyntheticTextEncoder, SyntheticVAE
            pipe.vae = pipe.vae or SyntheticVAE()
            pipe.text_encoder = pipe.text_encoder or SyntheticTextEncoder(dim=int(_first(cfg["cross_attention_dim"])))
        return pipe

    def to(self, device):
        self._device = torch.device(device)
        self.unet.to(self._device)
        for m in (self.vae, self.text_encoder):
            if m is not None and hasattr(m, "to"):
                m.to(self._device)
        return self

    def register_modules(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def device(self):
        return self._device

    @property
    def _execution_device(self):
        return self._device

    def progress_bar(self, iterable=None, total=None):
        return iterable

    @torch.no_grad()
    # memory savers of the reference pipeline (``pipeline_video_editing.py:229-245``): accepted, nothing to do on this device
    def enable_vae_slicing(self):
        self._vae_slicing = True

    def disable_vae_slicing(self):
        self._vae_slicing = False

    def enable_sequential_cpu_offload(self, gpu_id=0):
        logger.warning("enable_sequential_cpu_offload: the weights stay resident on the GPU (1250 M parameters of 288 GB)")

    def prepare_extra_step_kwargs(self, generator, eta):
        """``:373-388``: ``eta`` / ``generator`` for a scheduler whose ``step`` takes them."""
        import inspect
        params = set(inspect.signature(self.scheduler.step).parameters)
        return {k: v for k, v in (("eta", eta), ("generator", generator)) if k in params}

    def init_filter(self, video_length, height, width, filter_params):
        """``pipeline_video_editing.py:208-227``."""
        shape = [1, self.unet.config.in_channels, video_length, height // self.vae_scale_factor, width // self.vae_scale_factor]
        self.freq_filter = get_freq_filter(shape, device=self._execution_device, filter_type=filter_params.method,
                                           n=filter_params.n if filter_params.method == "butterworth" else None,
                                           d_s=filter_params.d_s, d_t=filter_params.d_t)

    def _frameinit(self, latents, clean, video_length, noise_level):
        """``:619-633``: the start noise's low frequencies replaced by those of the first frame, repeated over the clip and diffused to
        ``noise_level``."""
        if self.freq_filter is None:
            raise ValueError("use_frameinit needs init_filter(video_length, height, width, filter_params) first")
        static = self._motion_latents if getattr(self, "_motion_latents", None) is not None else clean.unsqueeze(2).repeat(1, 1, video_length, 1, 1)
        t = torch.full((latents.shape[0],), int(noise_level)).long()
        z_T = self.scheduler.add_noise(original_samples=static.to(latents.device), noise=latents, timesteps=t)
        return freq_mix_3d(z_T.to(torch.float32), latents, LPF=self.freq_filter).to(latents.dtype)

    # ------------------------------------------------------------------ checks / encoders
    def check_inputs(self, prompt, height, width, callback_steps=1, first_frame_paths=None):
        """``pipeline_video_editing.py:390-406``."""
        if not isinstance(prompt, str) and not isinstance(prompt, list):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if first_frame_paths is not None and (not isinstance(prompt, str) and not isinstance(first_frame_paths, list)):
            raise ValueError(f"`first_frame_paths` has to be of type `str` or `list` but is {type(first_frame_paths)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if (callback_steps is None) or (callback_steps is not None and (not isinstance(callback_steps, int) or callback_steps <= 0)):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")

    @staticmethod
    def _one(x, what):
        if isinstance(x, (list, tuple)):
            if len(x) != 1:
                raise NotImplementedError(f"one clip per call: got {len(x)} {what}")
            return x[0]
        return x

    @staticmethod
    def _guidance_mode(guidance_scale_txt, guidance_scale_img):
        """``:770-775``: None / "text" ([uncond, text]) / "both" ([uncond, image, image + text])."""
        mode = None
        if guidance_scale_txt > 1.0:
            mode = "text"
        if guidance_scale_img > 1.0:
            mode = "both"
        return mode

    def _encode_prompt(self, prompt, device, num_videos_per_prompt, do_classifier_free_guidance, negative_prompt):
        """``:261-351``: last hidden state of the text encoder (final LayerNorm applied), rows [uncond (, uncond), text]."""
        if num_videos_per_prompt != 1:
            raise NotImplementedError("num_videos_per_prompt > 1")
        prompt = self._one(prompt, "prompts")
        te = self.text_encoder.encode(prompt, device, None).to(torch.float16)
        if do_classifier_free_guidance:
            neg = "" if negative_prompt is None else self._one(negative_prompt, "negative prompts")
            if not isinstance(neg, str):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(neg)} != {type(prompt)}.")
            ue = self.text_encoder.encode(neg, device, None).to(torch.float16)
            te = torch.cat([ue, te] if do_classifier_free_guidance == "text" else [ue, ue, te])
        return te

    def _first_frame_latent(self, path_or_image, height, width, crop, device):
        """One first-frame image (path, PIL image, or -- ``first_frames`` -- an already pre-processed [1, 3, H, W] tensor) -> sampled,
        scaled VAE latent [1, 4, h, w] (``:796-833``)."""
        self._motion_latents = None
        if torch.is_tensor(path_or_image):
            return self.vae.encode_pixels(path_or_image.float(), device)
        img = path_or_image if isinstance(path_or_image, Image.Image) else Image.open(path_or_image).convert("RGB")
        motion, n = getattr(self, "_camera_motion", None), getattr(self, "_video_length", None)
        if motion is None:
            return self.vae.encode_pixels(frame_to_pixels(img, height, width, crop), device)
        # camera motion (``:553-577``): Resize(height) for a pan, Resize(2 * height) for a zoom, no centre crop; the pseudo clip's frames are
        # encoded together and its first latent is the conditioning frame, the whole of it the FrameInit layout
        x = (_resize(_to_tensor(img), height if motion.startswith("pan") else height * 2) - 0.5) / 0.5
        lat = self.vae.encode_pixels(camera_motion_frames(x, motion, n, width), device)        # [f, 4, h, w]
        self._motion_latents = lat.permute(1, 0, 2, 3)[None].contiguous()
        return lat[:1]

    def encode_vae_video(self, video: List[Image.Image], device, height: int = 576, width: int = 1024):
        """``:1226-1258``: every frame encoded on its own (one posterior sample per frame) -> [1, 4, F, h, w]."""
        lat = [self.vae.encode_pixels(frame_to_pixels(f, height, width, True), device)[0] for f in video]
        return torch.stack(lat).permute(1, 0, 2, 3)[None].contiguous()

    def decode_latents(self, latents, first_frames=None):
        """``:353-371``: frame-by-frame decode -> float32 numpy [b, c, f, H, W] in [0, 1]."""
        video = self.vae.decode_video(latents.to(torch.float16), decode_chunk_size=1)    # [1, 3, F, H, W] in [-1, 1]
        if first_frames is not None:
            video = torch.cat([first_frames.unsqueeze(2).to(video), video], dim=2)
        return (video / 2 + 0.5).clamp(0, 1).detach().cpu().float().numpy()

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator, latents=None,
                        noise_sampling_method="vanilla", noise_alpha=1.0):
        """``:408-466`` (one generator; the noise is drawn on the host so that a seed means the same latents on any device)."""
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            if isinstance(generator, list):
                raise NotImplementedError("a list of generators")
            r = lambda s: torch.randn(s, generator=generator, dtype=torch.float32)
            a2 = noise_alpha ** 2
            if noise_sampling_method == "vanilla":
                latents = r(shape)
            elif noise_sampling_method == "pyoco_mixed":
                base = r(shape[:2] + (1,) + shape[3:]) * math.sqrt(a2 / (1 + a2))
                latents = base + r(shape) * math.sqrt(1 / (1 + a2))
            elif noise_sampling_method == "pyoco_progressive":
                latents = r(shape)
                ind = r(shape) * math.sqrt(1 / (1 + a2))
                for j in range(1, video_length):
                    latents[:, :, j] = latents[:, :, j - 1] * math.sqrt(a2 / (1 + a2)) + ind[:, :, j]
            else:
                raise ValueError(noise_sampling_method)
        elif tuple(latents.shape) != shape:
            raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
        return latents.to(device=device, dtype=dtype) * self.scheduler.init_noise_sigma

    # ------------------------------------------------------------------ the loop
    def _common(self, prompt, height, width, callback_steps, first_frame_paths, first_frames, latents, num_videos_per_prompt, eta,
                guidance_rescale, use_frameinit, camera_motion):
        if first_frame_paths is not None and first_frames is not None:
            raise ValueError("Only one of `first_frame_paths` and `first_frames` can be passed.")
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps, first_frame_paths)
        if num_videos_per_prompt != 1:
            raise NotImplementedError("num_videos_per_prompt > 1 is not built (the AnyV2V runners leave it at 1)")
        self._camera_motion = camera_motion
        if first_frames is not None and (not torch.is_tensor(first_frames) or first_frames.dim() != 4 or first_frames.shape[0] != 1):
            raise NotImplementedError("first_frames: one pre-processed frame [1, 3, H, W] in [-1, 1] per call")
        if latents is not None and latents.shape[0] != 1:
            raise NotImplementedError(f"one clip per call: latents batch {latents.shape[0]}")
        return height, width

    def _denoise(self, latents, ff_input, text_embeddings, timesteps, frame_stride, branches, g_img, g_txt, source=None, on_step=None,
                 eta=0.0, generator=None, guidance_rescale=0.0):
        """``latents`` [1, C, F - 1, h, w]; ``ff_input`` [nb, C, 1, h, w]; ``branches`` = (b_unc, b_img, b_txt) of the guided rows;
        ``source(t)`` -> the source branch's latents at t (PnP: row 0 of the batch).

        One step = the UNet forward over all rows (static input buffers, ``_StepGraphs``) + one guided-step kernel.  The forward can
        be replayed as a HIP graph per injection state (ANYV2V_CONSISTI2V_GRAPHS=1); measured, that does not pay for this family.

        Two options no AnyV2V runner switches on: ``guidance_rescale`` (``:50-61,685-688``: the guided prediction rescaled towards the
        standard deviation of the text branch; a few torch reductions per step, then the step kernel on the combined prediction) and
        ``eta`` (``:373-388``: handed to a scheduler whose step takes it, i.e. the forward DDIM scheduler; the variance noise comes
        from ``generator``)."""
        nb = ff_input.shape[0]
        n_guided = nb - (1 if source is not None else 0)
        latents = latents.to(torch.float16).contiguous()
        pred = self.scheduler.prediction
        stochastic = eta != 0.0 and isinstance(self.scheduler, DDIMScheduler)   # (``prepare_extra_step_kwargs``: only a step that takes eta gets it)
        eng = self._step_graphs(nb, latents, text_embeddings, frame_stride)
        eng.ehs.copy_(text_embeddings)
        eng.ff.copy_(ff_input)
        for step_i, t in enumerate(timesteps):
            t = int(t)
            if source is not None:
                eng.x[0].copy_(source(t)[0], non_blocking=True)
                c2.register_time(self, t)
            for b in range(nb - n_guided, nb):
                eng.x[b].copy_(latents[0])
            eng.t_buf.fill_(float(t))
            e = eng.run(c2.injection_state(self) if source is not None else None)
            b_unc, b_img, b_txt, gi, gt = branches[0], branches[1], branches[2], g_img, g_txt
            if guidance_rescale > 0.0 and b_unc >= 0:
                if b_img >= 0:
                    raise NotImplementedError("guidance_rescale under image guidance (the reference reads an unbound name there: text guidance only)")
                eu, et = e[b_unc].float(), e[b_txt].float()
                cfg = eu + g_txt * (et - eu)
                cfg = cfg * (guidance_rescale * (et.std() / cfg.std()) + (1.0 - guidance_rescale))
                e, b_unc, b_txt, gt = cfg.to(torch.float16)[None].contiguous(), -1, 0, 1.0
            if stochastic:
                sa_t, sb_t, cx, ce, sigma = self.scheduler.eta_coefficients(t, eta)
                latents = ops.guided_step(e, latents, (sa_t, sb_t, cx, ce), b_unc=b_unc, b_img=b_img, b_txt=b_txt, g_img=gi, g_txt=gt,
                                          prediction=pred, noise=self.scheduler.draw_noise(latents, generator).contiguous(), sigma=sigma)
            else:
                latents = ops.guided_step(e, latents, self.scheduler.coefficients(t), b_unc=b_unc, b_img=b_img, b_txt=b_txt, g_img=gi,
                                          g_txt=gt, prediction=pred)
            if on_step is not None:
                on_step(step_i, t, latents)
        return latents

    def _step_graphs(self, nb, latents, ehs, frame_stride):
        if not self.unet._packed:
            self.unet.pack()
        key = (nb, tuple(latents.shape[1:]), tuple(ehs.shape[1:]), None if frame_stride is None else float(frame_stride), str(latents.device),
               id(self.unet), self.unet._pack_gen)
        eng = self._engines.pop(key, None)
        if eng is None:
            eng = _StepGraphs(self.unet, nb, latents, ehs, frame_stride)
        self._engines[key] = eng       # most recently used last
        while len(self._engines) > 3:  # (inversion / reconstruction, CFG sampling, PnP edit: every engine pins a graph pool)
            self._engines.pop(next(iter(self._engines)))
        return eng

    @staticmethod
    def _callback(callback, callback_steps):
        """``callback(i, t, latents)`` every ``callback_steps`` steps (``pipeline_video_editing.py:697-700``)."""
        if callback is None:
            return None
        return lambda i, t, x: callback(i, t, x) if i % (callback_steps or 1) == 0 else None

    @staticmethod
    def _branches(mode, offset=0):
        if mode is None:
            return (-1, -1, offset)
        if mode == "text":
            return (offset, -1, offset + 1)
        return (offset, offset + 1, offset + 2)

    def _ff_rows(self, mode, clean, noisy):
        """First-frame latent of every guided branch (``:896-903``): the image-unconditional branch of "both" sees the NOISY one."""
        if mode is None:
            return [clean]
        if mode == "text":
            return [clean, clean]
        return [noisy, clean, clean]

    def _finish(self, latents, first_frame_latents, output_type, return_dict):
        latents = torch.cat([first_frame_latents.unsqueeze(2).to(latents), latents], dim=2)
        if output_type == "latent":      # (extension: the reference returns latents from ``invert`` only)
            return AnimationPipelineOutput(videos=latents)
        video = self.decode_latents(latents)
        if output_type == "tensor":
            video = torch.from_numpy(video)
        if not return_dict:
            return video
        return AnimationPipelineOutput(videos=video)

    _crop_first_frame = False     # (``pipeline_video_editing.py:588-589``: plain Resize((height, width)); the animation pipelines crop)

    def _sample(self, clean, latents, text_embeddings, mode, video_length, height, width, num_inference_steps, t_idx, generator,
                noise_sampling_method, noise_alpha, use_frameinit, frameinit_noise_level, frame_stride, g_img, g_txt, on_step, eta=0.0,
                guidance_rescale=0.0):
        """Timesteps, start latents (given, or drawn; FrameInit), first-frame rows, the guided loop (``:603-704``) -> [1, C, F - 1, h, w]."""
        device = self._execution_device
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        self.scheduler.timesteps = self.scheduler.timesteps[t_idx:]
        latents = self.prepare_latents(1, self.unet.config.in_channels, video_length, height, width, torch.float16, device, generator, latents,
                                       noise_sampling_method, noise_alpha)
        if use_frameinit:
            latents = self._frameinit(latents, clean, video_length, frameinit_noise_level)
        noisy, latents = latents[:, :, 0], latents[:, :, 1:]
        ff = torch.cat(self._ff_rows(mode, clean, noisy)).unsqueeze(2)
        return self._denoise(latents, ff, text_embeddings, self.scheduler.timesteps, frame_stride, self._branches(mode), g_img, g_txt,
                             on_step=on_step, eta=eta, generator=generator, guidance_rescale=guidance_rescale)

    # ------------------------------------------------------------------ ``__call__`` (:469-711)
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]], video_length: Optional[int], height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale_txt: float = 7.5, guidance_scale_img: float = 2.0,
                 negative_prompt=None, num_videos_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents: Optional[torch.Tensor] = None, output_type: Optional[str] = "tensor", return_dict: bool = True, callback=None,
                 callback_steps: Optional[int] = 1, first_frame_paths=None, first_frames=None, noise_sampling_method: str = "vanilla",
                 noise_alpha: float = 1.0, guidance_rescale: float = 0.0, frame_stride: Optional[int] = None, use_frameinit: bool = False,
                 frameinit_noise_level: int = 999, camera_motion: str = None, ddim_init_latents_t_idx: Optional[int] = 0, **kwargs):
        height, width = self._common(prompt, height, width, callback_steps, first_frame_paths, first_frames, latents, num_videos_per_prompt,
                                     eta, guidance_rescale, use_frameinit, camera_motion)
        self._video_length = video_length
        device = self._execution_device
        mode = self._guidance_mode(guidance_scale_txt, guidance_scale_img)
        c2.clear_time(self)
        text_embeddings = self._encode_prompt(prompt, device, num_videos_per_prompt, mode, negative_prompt)
        if first_frame_paths is None and first_frames is None:
            raise NotImplementedError("sampling without a first frame (first_frame_condition_mode 'none')")
        clean = self._first_frame_latent(first_frames if first_frame_paths is None else self._one(first_frame_paths, "first frames"),
                                         height, width, self._crop_first_frame, device)
        latents = self._sample(clean, latents, text_embeddings, mode, video_length, height, width, num_inference_steps, ddim_init_latents_t_idx,
                               generator, noise_sampling_method, noise_alpha, use_frameinit, frameinit_noise_level, frame_stride,
                               guidance_scale_img, guidance_scale_txt, self._callback(callback, callback_steps), eta, guidance_rescale)
        return self._finish(latents, clean, output_type, return_dict)

    def sample_with_saving_features(self, *args, ddim_inv_latents_path=None, ddim_inv_prompt=None, ddim_inv_1st_frame_path=None, **kwargs):
        """``:971-1223``: ``__call__`` that also stamps the current timestep on the decoder's modules every step (for feature-saving
        hooks that are not in the reference tree; nothing reads the stamps unless an injection schedule is registered, and then
        ``sample_with_pnp`` is the entry point).  The three extra arguments of its signature are not used by its body either."""
        return self.__call__(*args, **kwargs)

    # ------------------------------------------------------------------ ``invert`` (:715-968)
    @torch.no_grad()
    def invert(self, prompt: Union[str, List[str]], video_length: Optional[int], height: Optional[int] = None, width: Optional[int] = None,
               num_inference_steps: int = 50, guidance_scale_txt: float = 7.5, guidance_scale_img: float = 2.0, negative_prompt=None,
               num_videos_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None,
               output_type: Optional[str] = "tensor", return_dict: bool = True, callback=None, callback_steps: Optional[int] = 1,
               first_frame_paths=None, first_frames=None, noise_sampling_method: str = "pyoco_mixed", noise_alpha: float = 1.0,
               guidance_rescale: float = 0.0, frame_stride: Optional[int] = None, use_frameinit: bool = False,
               frameinit_noise_level: int = 999, camera_motion: str = None, output_dir: Optional[str] = None,
               return_trajectory: bool = False, background_save: bool = False, **kwargs):
        """With ``self.scheduler`` an inverse scheduler: clean video latents -> noise, every step's latents (the clean first-frame
        latent as frame 0) written to ``output_dir/ddim_latents_{t}.pt``.  ``videos``: [1, n_steps, C, F, h, w], noisiest first."""
        height, width = self._common(prompt, height, width, callback_steps, first_frame_paths, first_frames, latents, num_videos_per_prompt,
                                     eta, guidance_rescale, use_frameinit, camera_motion)
        self._video_length = video_length
        device = self._execution_device
        mode = self._guidance_mode(guidance_scale_txt, guidance_scale_img)
        c2.clear_time(self)
        text_embeddings = self._encode_prompt(prompt, device, num_videos_per_prompt, mode, negative_prompt)
        if first_frame_paths is None and first_frames is None:
            raise NotImplementedError("inversion without a first frame (first_frame_condition_mode 'none')")
        clean = self._first_frame_latent(first_frames if first_frame_paths is None else self._one(first_frame_paths, "first frames"),
                                         height, width, True, device)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        latents = self.prepare_latents(1, self.unet.config.in_channels, video_length, height, width, torch.float16, device, generator, latents,
                                       noise_sampling_method, noise_alpha)
        if use_frameinit:
            latents = self._frameinit(latents, clean, video_length, frameinit_noise_level)
        noisy, latents = latents[:, :, 0], latents[:, :, 1:]
        ff = torch.cat(self._ff_rows(mode, clean, noisy)).unsqueeze(2)
        traj = LatentTrajectory()
        first = clean.unsqueeze(2).to(torch.float16)

        cb = self._callback(callback, callback_steps)

        def keep(i, t, x):
            traj[t] = torch.cat([first, x], dim=2)
            if cb is not None:
                cb(i, t, x)
        latents = self._denoise(latents, ff, text_embeddings, self.scheduler.timesteps, frame_stride, self._branches(mode),
                                guidance_scale_img, guidance_scale_txt, on_step=keep, guidance_rescale=guidance_rescale)
        ts = [int(t) for t in self.scheduler.timesteps]
        if output_dir is not None:
            traj.save(output_dir, background=background_save)
            logger.info(f"saved noisy latents for {len(ts)} timesteps to {output_dir}")
        self._last_trajectory = traj
        if return_trajectory:
            return traj
        inverted = torch.stack([traj[t] for t in reversed(ts)], 1)
        if output_type == "latent":
            return AnimationPipelineOutput(videos=inverted)
        video = self.decode_latents(latents)
        if output_type == "tensor":
            video = torch.from_numpy(video)
        if not return_dict:
            return video
        return AnimationPipelineOutput(videos=video)

    # ------------------------------------------------------------------ ``sample_with_pnp`` (:1261-1576)
    @torch.no_grad()
    def sample_with_pnp(self, prompt: Union[str, List[str]], video_length: Optional[int], height: Optional[int] = None,
                        width: Optional[int] = None, num_inference_steps: int = 50, guidance_scale_txt: float = 7.5,
                        guidance_scale_img: float = 2.0, negative_prompt=None, num_videos_per_prompt: Optional[int] = 1, eta: float = 0.0,
                        generator=None, latents: Optional[torch.Tensor] = None, output_type: Optional[str] = "tensor",
                        return_dict: bool = True, callback=None, callback_steps: Optional[int] = 1, first_frame_paths=None,
                        first_frames=None, noise_sampling_method: str = "vanilla", noise_alpha: float = 1.0, guidance_rescale: float = 0.0,
                        frame_stride: Optional[int] = None, use_frameinit: bool = False, frameinit_noise_level: int = 999,
                        camera_motion: str = None, ddim_init_latents_t_idx: Optional[int] = 0, ddim_inv_latents_path=None,
                        ddim_inv_prompt: Union[str, List[str]] = None, ddim_inv_1st_frame_path=None, **kwargs):
        """Batch rows [source (inversion prompt, stored latents at t) | negative, editing]; the hooks registered with
        ``anyv2v_amd.consisti2v.register_*`` copy the source row's features into the other two on their schedules."""
        height, width = self._common(prompt, height, width, callback_steps, first_frame_paths, first_frames, latents, num_videos_per_prompt,
                                     eta, guidance_rescale, use_frameinit, camera_motion)
        self._video_length = video_length
        device = self._execution_device
        mode = self._guidance_mode(guidance_scale_txt, guidance_scale_img)
        if mode != "text":
            raise NotImplementedError("sample_with_pnp needs text guidance only (guidance_scale_txt > 1, guidance_scale_img <= 1): the hooks "
                                      "split the batch in [source, negative, editing] (consisti2v/pnp_utils.py:96,188,296)")
        text_embeddings = self._encode_prompt(prompt, device, num_videos_per_prompt, mode, negative_prompt)
        src_embeds = self._encode_prompt(ddim_inv_prompt, device, num_videos_per_prompt, None, None)
        text_embeddings = torch.cat([src_embeds, text_embeddings])
        if first_frame_paths is None or ddim_inv_1st_frame_path is None:
            raise ValueError("sample_with_pnp needs first_frame_paths (the edited first frame) and ddim_inv_1st_frame_path")
        edited = self._one(first_frame_paths, "first frames")
        clean = self._first_frame_latent(edited, height, width, False, device)
        # the reference opens ddim_inv_1st_frame_path and then encodes the EDITED frame again (:1426): a second posterior sample of it
        self._one(ddim_inv_1st_frame_path, "inversion first frames")
        src_first = self._first_frame_latent(edited, height, width, False, device)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        self.scheduler.timesteps = self.scheduler.timesteps[ddim_init_latents_t_idx:]
        latents = self.prepare_latents(1, self.unet.config.in_channels, video_length, height, width, torch.float16, device, generator, latents,
                                       noise_sampling_method, noise_alpha)
        if use_frameinit:
            latents = self._frameinit(latents, clean, video_length, frameinit_noise_level)
        noisy, latents = latents[:, :, 0], latents[:, :, 1:]
        ff = torch.cat([src_first] + self._ff_rows(mode, clean, noisy)).unsqueeze(2)

        def source(t):
            return load_ddim_latents_at_t(t, ddim_inv_latents_path).to(device=device, dtype=torch.float16)[:, :, 1:]
        try:
            latents = self._denoise(latents, ff, text_embeddings, self.scheduler.timesteps, frame_stride, self._branches(mode, 1),
                                    guidance_scale_img, guidance_scale_txt, source=source,
                                    on_step=self._callback(callback, callback_steps), eta=eta, generator=generator,
                                    guidance_rescale=guidance_rescale)
        finally:
            c2.clear_time(self)
        return self._finish(latents, clean, output_type, return_dict)


class ConditionalAnimationPipeline(ConditionalVideoEditingPipeline):
    """ConsistI2V's image-to-video sampler (``consisti2v/consisti2v/pipelines/pipeline_conditional_animation.py:462-703``): the editing
    pipeline's ``__call__`` with the first frame pre-processed as Resize(height) + CenterCrop((height, width)) (``:539-540``) and always
    from the first timestep.  It has no ``invert`` / ``sample_with_pnp``."""
    _crop_first_frame = True

    def __call__(self, *args, **kwargs):
        if kwargs.pop("ddim_init_latents_t_idx", 0):
            raise TypeError("ConditionalAnimationPipeline samples from the first timestep (ddim_init_latents_t_idx is the editing pipeline's)")
        return super().__call__(*args, **kwargs)

    def invert(self, *args, **kwargs):
        raise AttributeError("invert is ConditionalVideoEditingPipeline's")

    def sample_with_pnp(self, *args, **kwargs):
        raise AttributeError("sample_with_pnp is ConditionalVideoEditingPipeline's")


class AutoregressiveAnimationPipeline(ConditionalAnimationPipeline):
    """Long clips, chunk by chunk (``pipeline_autoregress_animation.py:401-615``): ``autoregress_steps`` samplings of ``video_length``
    frames, each from fresh noise of the one generator, each conditioned on (and, with FrameInit, laid out after) the LAST latent frame
    of the chunk before; chunks overlap by that frame, so the result has ``video_length * n - n + 1`` frames, decoded in one go."""

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]], video_length: Optional[int], height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale_txt: float = 7.5, guidance_scale_img: float = 2.0,
                 negative_prompt=None, num_videos_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents: Optional[torch.Tensor] = None, output_type: Optional[str] = "tensor", return_dict: bool = True, callback=None,
                 callback_steps: Optional[int] = 1, first_frame_paths=None, first_frames=None, noise_sampling_method: str = "vanilla",
                 noise_alpha: float = 1.0, guidance_rescale: float = 0.0, frame_stride: Optional[int] = None, autoregress_steps: int = 3,
                 use_frameinit: bool = False, frameinit_noise_level: int = 999, **kwargs):
        if kwargs.get("ddim_init_latents_t_idx", 0) or kwargs.get("camera_motion") is not None:
            raise TypeError("AutoregressiveAnimationPipeline has no ddim_init_latents_t_idx / camera_motion (the reference's **kwargs would drop them)")
        height, width = self._common(prompt, height, width, callback_steps, first_frame_paths, first_frames, latents, num_videos_per_prompt,
                                     eta, guidance_rescale, use_frameinit, None)
        self._video_length = video_length
        device = self._execution_device
        mode = self._guidance_mode(guidance_scale_txt, guidance_scale_img)
        c2.clear_time(self)
        text_embeddings = self._encode_prompt(prompt, device, num_videos_per_prompt, mode, negative_prompt)
        if first_frame_paths is None and first_frames is None:
            raise NotImplementedError("sampling without a first frame (first_frame_condition_mode 'none')")
        clean = self._first_frame_latent(first_frames if first_frame_paths is None else self._one(first_frame_paths, "first frames"),
                                         height, width, True, device)
        chunks = [clean.unsqueeze(2).to(torch.float16)]
        for _ in range(int(autoregress_steps)):
            x = self._sample(clean, latents, text_embeddings, mode, video_length, height, width, num_inference_steps, 0, generator,
                             noise_sampling_method, noise_alpha, use_frameinit, frameinit_noise_level, frame_stride, guidance_scale_img,
                             guidance_scale_txt, self._callback(callback, callback_steps), eta, guidance_rescale)
            chunks.append(x)
            clean, latents = x[:, :, -1].to(clean.dtype), None     # (``:599-603``: given start latents serve the first chunk only)
        full = torch.cat(chunks, dim=2)
        return self._finish(full[:, :, 1:], full[:, :, 0], output_type, return_dict)


def _first(v):
    return v[0] if isinstance(v, (list, tuple)) else v


def _sched_defaults(root):
    """The ConsistI2V scheduler configuration unless a local ``scheduler_config.json`` says otherwise."""
    return {} if os.path.isfile(os.path.join(root, "scheduler", "scheduler_config.json")) else dict(CONSISTI2V_SCHEDULER_CONFIG)


def inverse_scheduler_from_pretrained(root, subfolder="scheduler"):
    return DDIMInverseScheduler.from_pretrained(str(root), subfolder=subfolder, **_sched_defaults(str(root)))


def init_random_weights_(unet: c2.VideoLDMUNet3DConditionModel, seed: int):
    """Deterministic weights of a plausible scale for runs without a checkpoint (1 / sqrt(fan_in); norms near identity; blend
    factors 0.5 so that the temporal layers take part)."""
    g = torch.Generator().manual_seed(seed)
    sd = unet.state_dict()
    new = {}
    for name in sorted(sd):
        v = sd[name]
        if name.endswith("freqs"):
            new[name] = v.clone()
        elif name.endswith("alpha"):
            new[name] = torch.full_like(v, 0.5)
        elif v.dim() >= 2:
            new[name] = torch.randn(v.shape, generator=g) * (1.0 / v[0].numel() ** 0.5)
        elif "norm" in name and name.endswith("weight"):
            new[name] = torch.ones_like(v)
        else:
            new[name] = torch.zeros_like(v)
    unet.load_state_dict(new)
    return unet
