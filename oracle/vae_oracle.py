"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch, fp32) of diffusers-0.26.3 ``AutoencoderKL`` as configured
for the SD-2.x VAE that ``ali-vilab/i2vgen-xl`` ships (4 latent channels, block_out_channels
(128, 256, 512, 512), layers_per_block 2, 32 norm groups, scaling_factor 0.18215).

Used by the pipeline stages either side of the hot loop: ``encode_vae_video`` (``i2vgen-xl/pipelines/
pipeline_i2vgen_xl.py:565-592``), ``prepare_image_latents`` (``:532-562``) and ``decode_latents`` (``:598-620``).
Only ``tests/`` and ``__graft_entry__.smoke()`` may import this module; the product path is ``anyv2v_amd/vae.py``.

PARITY STATUS: **unpinned**.  ``diffusers`` is not installed and the reference tree vendors no VAE code or fixture, so
this file restates the published architecture (diffusers ``models/autoencoders/vae.py``, ``models/resnet.py``,
``models/attention_processor.py`` at tag v0.26.3) from its documented structure: state-dict key names follow the
checkpoint layout (``encoder.down_blocks.{i}.resnets.{j}.norm1.weight`` ...), so real weights load into both this
oracle and the native module.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215

    @staticmethod
    def mini():
        return VAEConfig(block_out_channels=(64, 128), layers_per_block=1)


class ResnetBlock(nn.Module):
    """diffusers ResnetBlock2D with temb_channels=None, eps 1e-6, output_scale_factor 1."""

    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class Attention(nn.Module):
    """Single head of width C over the H*W tokens (heads = C // attention_head_dim with attention_head_dim = C),
    residual connection, group_norm in front, rescale_output_factor 1."""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Identity()])

    def forward(self, x):
        n, c, h, w = x.shape
        t = self.group_norm(x).view(n, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        o = self.to_out[0](o).transpose(1, 2).reshape(n, c, h, w)
        return o + x


class MidBlock(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(c, c, groups), ResnetBlock(c, c, groups)])
        self.attentions = nn.ModuleList([Attention(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Downsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class Upsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample(cout)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.downsamplers is None else self.downsamplers[0](x)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample(cout)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)


class Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, c in enumerate(boc):
            cin, out = out, c
            self.down_blocks.append(DownBlock(cin, out, cfg.layers_per_block, g, i != len(boc) - 1))
        self.mid_block = MidBlock(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        rev = list(reversed(boc))
        self.conv_in = nn.Conv2d(cfg.latent_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlock(rev[0], g)
        self.up_blocks = nn.ModuleList()
        out = rev[0]
        for i, c in enumerate(rev):
            cin, out = out, c
            self.up_blocks.append(UpBlock(cin, out, cfg.layers_per_block + 1, g, i != len(boc) - 1))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKLOracle(nn.Module):
    def __init__(self, cfg: VAEConfig = VAEConfig()):
        super().__init__()
        self.cfg = cfg
        self.encoder, self.decoder = Encoder(cfg), Decoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)

    @torch.no_grad()
    def encode_moments(self, x):
        """[n,3,H,W] in [-1,1] -> (mean, logvar) of the diagonal Gaussian posterior, each [n,4,H/8,W/8]."""
        m = self.quant_conv(self.encoder(x))
        mean, logvar = m.chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)

    @torch.no_grad()
    def decode(self, z):
        """[n,4,h,w] (already divided by scaling_factor) -> [n,3,8h,8w]."""
        return self.decoder(self.post_quant_conv(z))


def random_state_dict(cfg: VAEConfig, seed: int):
    """Seeded random weights rounded to fp16 (shared by the oracle and the native module in the parity tests)."""
    torch.manual_seed(seed)
    m = AutoencoderKLOracle(cfg)
    sd = {}
    for k, v in m.state_dict().items():
        if v.dim() >= 2:
            fan_in = v[0].numel()
            w = torch.randn_like(v) / fan_in ** 0.5
        elif k.endswith("weight"):
            w = 1.0 + 0.1 * torch.randn_like(v)   # norm gains
        else:
            w = 0.05 * torch.randn_like(v)
        sd[k] = w.half().float()
    return sd
