"""Encoders either side of the hot loop -- component interface + offline stand-ins.

The reference wires ``AutoencoderKL`` / ``CLIPTextModel`` / ``CLIPVisionModelWithProjection`` from the hub repo
``ali-vilab/i2vgen-xl`` (``pipeline_i2vgen_xl.py:155-178``).  Neither the weights nor ``diffusers`` exist offline, and
these once-per-clip stages are outside the HIP hot path.  The pipeline only needs four small interfaces, defined here.
``NativeVAE`` is the real AutoencoderKL architecture on the HIP kernels (``anyv2v_amd/vae.py``, diffusers state-dict
keys, random weights offline).  The text / image encoders and ``SyntheticVAE`` are *synthetic* implementations
(deterministic, weight-free, NOT the real models) so that the CLIs, file formats and multi-GPU sharding can be
exercised end to end without any checkpoint:

  vae.encode_image(pil, device, height, width) -> [1,4,h,w]      (posterior mean x scaling_factor)
  vae.encode_video(list[pil], device, height, width) -> [1,4,F,h,w]
  vae.decode_video(latents[1,4,F,h,w], decode_chunk_size) -> float32 [1,3,F,H,W] in [-1,1] (host or device);  vae.to_pil(video)
  text_encoder.encode(list[str], device, clip_skip) -> [n,77,1024]
  image_encoder.encode(pil, width, device) -> [1,1,1024]
"""
from __future__ import annotations

import hashlib
from types import SimpleNamespace
from typing import List

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image


def _center_crop_wide(image: Image.Image, resolution):
    """``pipeline_i2vgen_xl.py:1487-1509``: resize to cover, then centre-crop to ``resolution`` = (w, h).  The resized
    size is ``round(width // scale)`` -- a FLOOR division, as the reference writes it (``:1496,1505``)."""
    w, h = image.size
    scale = min(w / resolution[0], h / resolution[1])
    image = image.resize((round(w // scale), round(h // scale)), resample=Image.BOX)
    x1 = (image.width - resolution[0]) // 2
    y1 = (image.height - resolution[1]) // 2
    return image.crop((x1, y1, x1 + resolution[0], y1 + resolution[1]))


def _pil_to_tensor(img: Image.Image) -> torch.Tensor:
    a = np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1)[None] * 2.0 - 1.0  # [1,3,H,W] in [-1,1]


_U8_LUT = None


def _pil_batch_to_device(frames: List[Image.Image], device) -> torch.Tensor:
    """``torch.cat([_pil_to_tensor(f) for f in frames])`` with the float conversion done on ``device``: the frames cross the
    bus as uint8 (a quarter of the bytes) and are mapped through a 256-entry table that holds exactly the values
    ``_pil_to_tensor`` computes on the host, so the result is bit-identical to the host path."""
    global _U8_LUT
    if _U8_LUT is None:
        _U8_LUT = torch.from_numpy(np.arange(256, dtype=np.float32) / 255.0) * 2.0 - 1.0
    u8 = torch.from_numpy(np.stack([np.asarray(f.convert("RGB"), dtype=np.uint8) for f in frames]))  # [F,H,W,3]
    return _U8_LUT.to(device)[u8.to(device).long()].permute(0, 3, 1, 2).contiguous()


class SyntheticVAE:
    """8x area-pool + fixed orthogonal-ish 3->4 channel mix (and its pseudo-inverse for decoding).  Weight-free
    stand-in with the SD-VAE's shapes / scaling factor; deterministic (the real VAE *samples* its posterior with the
    global RNG, ``pipeline_i2vgen_xl.py:540,582``)."""

    def __init__(self):
        self.config = SimpleNamespace(scaling_factor=0.18215, block_out_channels=(128, 256, 512, 512))
        g = torch.Generator().manual_seed(20240229)
        self.mix = torch.linalg.qr(torch.randn(4, 4, generator=g))[0][:, :3].contiguous()  # [4,3]

    def to(self, device):
        return self

    def _enc(self, x):  # [n,3,H,W] -> [n,4,H/8,W/8]
        x = F.avg_pool2d(x, 8)
        return torch.einsum("oc,nchw->nohw", self.mix, x) * (self.config.scaling_factor * 4.0)

    def encode_pixels(self, x, device):
        """[n, 3, H, W] in [-1, 1] -> scaled latents [n, 4, H/8, W/8] fp16 (callers that pre-process frames their own way)."""
        return self._enc(x.float().cpu()).to(device=device, dtype=torch.float16)

    def encode_image(self, image, device, height, width):
        x = _pil_to_tensor(_center_crop_wide(image, (width, height)))
        return self._enc(x).to(device=device, dtype=torch.float16)

    def encode_video(self, video: List[Image.Image], device, height, width):
        lat = torch.cat([self._enc(_pil_to_tensor(_center_crop_wide(f, (width, height)))) for f in video])  # [F,4,h,w]
        return lat.permute(1, 0, 2, 3)[None].to(device=device, dtype=torch.float16)

    def decode_video(self, latents, decode_chunk_size=None):
        z = latents.float().cpu() / (self.config.scaling_factor * 4.0)
        b, c, f, h, w = z.shape
        x = torch.einsum("oc,bofhw->bcfhw", self.mix, z)
        x = F.interpolate(x.reshape(b, 3 * f, h, w), scale_factor=8, mode="nearest").reshape(b, 3, f, h * 8, w * 8)
        return x.clamp(-1, 1)

    def to_pil(self, video):
        """``tensor2vid`` (``pipeline_i2vgen_xl.py:79-97``) for one clip: [1,3,F,H,W] in [-1,1] -> list of PIL.  The
        quantisation runs where the tensor lives (same fp32 arithmetic either way); a device tensor crosses the bus as uint8."""
        x = ((video[0].permute(1, 2, 3, 0) + 1.0) * 127.5).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
        return [Image.fromarray(fr) for fr in x]


class NativeVAE:
    """The real AutoencoderKL architecture on the HIP kernels (``anyv2v_amd/vae.py``) behind the pipeline's VAE
    interface: ``encode_vae_video`` / ``prepare_image_latents`` sample the posterior with the global RNG and scale by
    ``scaling_factor`` (``pipeline_i2vgen_xl.py:540,582``); ``decode_latents`` divides by it and decodes
    ``decode_chunk_size`` frames at a time (``:598-620``).  Weights: ``vae.load_state_dict(diffusers_state_dict)``;
    offline only seeded random weights are available (``random_init_seed``)."""

    def __init__(self, state_dict=None, random_init_seed=None, cfg=None, sample_posterior=True):
        from .vae import AutoencoderKL, init_random_weights_
        self.model = AutoencoderKL(cfg)
        if state_dict is not None:
            self.model.load_state_dict(state_dict)
        else:
            init_random_weights_(self.model, 0 if random_init_seed is None else random_init_seed)
        self.config = SimpleNamespace(scaling_factor=self.model.cfg.scaling_factor,
                                      block_out_channels=self.model.cfg.block_out_channels)
        self.sample_posterior = sample_posterior

    def to(self, device):
        self.model.to(device)
        return self

    def _encode(self, x, device):
        self.model.to(device)
        mean, logvar = self.model.encode_moments(x)
        z = mean + torch.exp(0.5 * logvar) * torch.randn_like(mean) if self.sample_posterior else mean
        return z * self.config.scaling_factor

    def encode_pixels(self, x, device):
        """[n, 3, H, W] in [-1, 1] -> sampled, scaled latents [n, 4, H/8, W/8] fp16."""
        return self._encode(x.to(device), device).to(torch.float16)

    def encode_image(self, image, device, height, width):
        x = _pil_batch_to_device([_center_crop_wide(image, (width, height))], device)
        return self._encode(x, device).to(torch.float16)

    def encode_video(self, video: List[Image.Image], device, height, width):
        x = _pil_batch_to_device([_center_crop_wide(f, (width, height)) for f in video], device)  # [F,3,H,W]
        return self._encode(x, device).permute(1, 0, 2, 3)[None].to(torch.float16)

    def decode_video(self, latents, decode_chunk_size=None):
        """-> float32 [1,3,F,H,W] in [-1,1] on the latents' device (as the reference's ``decode_latents`` leaves it)."""
        z = latents[0].permute(1, 0, 2, 3).float() / self.config.scaling_factor  # [F,4,h,w]
        n = z.shape[0]
        chunk = n if not decode_chunk_size else int(decode_chunk_size)
        frames = [self.model.decode(z[i:i + chunk]) for i in range(0, n, chunk)]
        return torch.cat(frames).permute(1, 0, 2, 3)[None].float().clamp(-1, 1)  # [1,3,F,H,W]

    to_pil = SyntheticVAE.to_pil


def _seeded(text: str, shape):
    seed = int.from_bytes(hashlib.sha256(text.encode()).digest()[:8], "little") % (2 ** 63)
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


class SyntheticTextEncoder:
    """Hash-seeded N(0,1) token embeddings with the OpenCLIP ViT-H text tower's output shape (77 x 1024)."""

    def __init__(self, dim=1024, tokens=77):
        self.dim, self.tokens = dim, tokens

    def to(self, device):
        return self

    def encode(self, prompts, device, clip_skip=None):
        if isinstance(prompts, str):
            prompts = [prompts]
        e = torch.stack([_seeded("txt:" + p, (self.tokens, self.dim)) for p in prompts])
        return e.to(device=device, dtype=torch.float16)


class SyntheticImageEncoder:
    """Image embedding = fixed random projection of the 224x224 bilinear-resized, centre-cropped frame."""

    def __init__(self, dim=1024):
        self.dim = dim
        self.proj = torch.randn(dim, 3 * 14 * 14, generator=torch.Generator().manual_seed(7)) / (3 * 14 * 14) ** 0.5

    def to(self, device):
        return self

    def encode(self, image, width, device):
        img = _center_crop_wide(image, (width, width)).resize((224, 224), resample=Image.BILINEAR)
        x = F.avg_pool2d(_pil_to_tensor(img), 16).reshape(-1)
        return (self.proj @ x)[None, None].to(device=device, dtype=torch.float16)


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class NativeTextEncoder:
    """The checkpoint's CLIP text tower on the HIP kernels (``anyv2v_amd.clip.CLIPTextTower``) behind ``text_encoder.encode``:
    ``encode_prompt`` of the reference (``pipeline_i2vgen_xl.py:224-409``) -- pad / truncate to ``model_max_length``, and with
    ``clip_skip`` hidden state ``-(clip_skip + 1)`` followed by ``final_layer_norm`` (``:312-324``).  The tokenizer (host-side BPE
    string processing) is ``transformers.CLIPTokenizer``."""

    def __init__(self, tower, tokenizer):
        self.tower, self.tokenizer = tower, tokenizer

    def to(self, device):
        self.tower.to(device)
        return self

    @torch.no_grad()
    def encode(self, prompts, device, clip_skip=None):
        if isinstance(prompts, str):
            prompts = [prompts]
        ids = self.tokenizer(prompts, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                             return_tensors="pt").input_ids
        return self.tower.ensure(device).encode_ids(ids, clip_skip)


class NativeImageEncoder:
    """The checkpoint's CLIP vision tower + projection on the HIP kernels (``anyv2v_amd.clip.CLIPVisionTower``) behind
    ``image_encoder.encode`` -- ``_encode_image`` (``pipeline_i2vgen_xl.py:411-441``); same pre-processing as tests/hf_clip_reference.py::HFImageEncoder."""

    def __init__(self, tower, crop=224):
        self.tower, self.crop = tower, crop

    def to(self, device):
        self.tower.to(device)
        return self

    @torch.no_grad()
    def encode(self, image, width, device):
        img = _center_crop_wide(image, (width, width)).resize((self.crop, self.crop), resample=Image.BILINEAR)
        x = (_pil_to_tensor(img) + 1.0) * 0.5  # [1,3,224,224] in [0,1]
        mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
        std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
        x = (x - mean) / std
        return self.tower.ensure(device).image_embeds(x)[:, None]  # [1,1,1024]


def _load_tower_files(folder: str):
    """(config dict, state dict) of one transformers model folder: config.json + model(.fp16).safetensors / pytorch_model.bin."""
    import json
    import os
    cfg = json.load(open(os.path.join(folder, "config.json")))
    for name in ("model.fp16.safetensors", "model.safetensors"):
        f = os.path.join(folder, name)
        if os.path.isfile(f):
            from safetensors.torch import load_file
            return cfg, load_file(f)
    for name in ("pytorch_model.fp16.bin", "pytorch_model.bin"):
        f = os.path.join(folder, name)
        if os.path.isfile(f):
            return cfg, torch.load(f, map_location="cpu")
    raise FileNotFoundError(f"no weights file in {folder}")


def attach_native_text_encoder(pipe, root: str) -> bool:
    """``<root>/text_encoder`` + ``<root>/tokenizer`` alone (Stable-Diffusion-style checkpoints have no image encoder: the ConsistI2V
    and SEINE backends).  Returns True when attached."""
    import os
    from .clip import CLIPTextTower, CLIPTowerConfig
    te, tk = os.path.join(root, "text_encoder"), os.path.join(root, "tokenizer")
    if not (os.path.isdir(te) and os.path.isdir(tk)):
        return False
    from transformers import CLIPTokenizer
    pipe.tokenizer = CLIPTokenizer.from_pretrained(tk)
    tcfg, tsd = _load_tower_files(te)
    pipe.text_encoder = NativeTextEncoder(CLIPTextTower(CLIPTowerConfig.from_hf(tcfg, "text"), tsd), pipe.tokenizer)
    return True


def attach_native_clip_encoders(pipe, root: str):
    """Load ``<root>/text_encoder``, ``<root>/tokenizer``, ``<root>/image_encoder`` (the sub-folders of the
    ``ali-vilab/i2vgen-xl`` checkpoint) onto the native towers when they exist locally.  Returns True when attached."""
    import os
    from .clip import CLIPTextTower, CLIPTowerConfig, CLIPVisionTower
    need = [os.path.join(root, d) for d in ("text_encoder", "tokenizer", "image_encoder")]
    if not all(os.path.isdir(d) for d in need):
        return False
    from transformers import CLIPTokenizer
    pipe.tokenizer = CLIPTokenizer.from_pretrained(need[1])
    tcfg, tsd = _load_tower_files(need[0])
    vcfg, vsd = _load_tower_files(need[2])
    pipe.text_encoder = NativeTextEncoder(CLIPTextTower(CLIPTowerConfig.from_hf(tcfg, "text"), tsd), pipe.tokenizer)
    vc = CLIPTowerConfig.from_hf(vcfg, "vision")
    if vc.projection_dim is None:
        vc.projection_dim = vcfg.get("projection_dim")
    pipe.image_encoder = NativeImageEncoder(CLIPVisionTower(vc, vsd), crop=vc.image_size or 224)
    pipe.feature_extractor = object()
    return True


def attach_native_vae(pipe, state_dict=None, random_init_seed=0, cfg=None):
    """Swap the VAE stand-in for the native AutoencoderKL (real architecture; real weights if a state dict is given)."""
    pipe.vae = NativeVAE(state_dict, random_init_seed, cfg)
    return pipe


def attach_synthetic_encoders(pipe):
    pipe.vae = SyntheticVAE()
    pipe.text_encoder = SyntheticTextEncoder(pipe.unet.cfg.cross_attention_dim)
    pipe.tokenizer = object()
    pipe.image_encoder = SyntheticImageEncoder(pipe.unet.cfg.cross_attention_dim)
    pipe.feature_extractor = object()
    return pipe
