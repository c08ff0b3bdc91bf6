"""Native AutoencoderKL (the SD-2.x VAE of ``ali-vilab/i2vgen-xl``) on the HIP kernels -- SURVEY.md 8(f) F1.

Replaces ``self.vae`` of the reference pipeline for ``encode_vae_video`` (``i2vgen-xl/pipelines/
pipeline_i2vgen_xl.py:565-592``), ``prepare_image_latents`` (``:532-562``) and ``decode_latents`` (``:598-620``).
Same state-dict keys as diffusers-0.26.3 ``AutoencoderKL`` (``encoder.down_blocks.0.resnets.0.norm1.weight`` ...), so a
real checkpoint loads with ``load_state_dict``; offline only seeded random weights exist (parity vs
``oracle/vae_oracle.py``, itself unpinned -- see its header).

Layout and kernels are those of the UNet: channels-last token matrices ``[(n)(h w), C]`` fp16; every 3x3 convolution is
the implicit-GEMM kernel (stride-2 downsample with the encoder's one-sided padding = ``asym``, nearest x2 upsample folded
into the gather, residual / shortcut fused into the epilogue), GroupNorm(+SiLU) is the two-kernel GroupNorm, and the
single 512-wide attention head of the mid block runs as logits GEMM (fp32 out) -> row softmax -> value GEMM.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_F32OUT
from .unet import Conv2d, GroupNorm, Linear

PAD_CIN = 64  # 3 (RGB) / 4 (latent) input channels are zero-padded to one MFMA K-tile


class VAEConfig:
    def __init__(self, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215):
        self.in_channels, self.out_channels, self.latent_channels = in_channels, out_channels, latent_channels
        self.block_out_channels, self.layers_per_block = tuple(block_out_channels), layers_per_block
        self.norm_num_groups, self.scaling_factor = norm_num_groups, scaling_factor

    @staticmethod
    def mini():
        return VAEConfig(block_out_channels=(64, 128), layers_per_block=1)


class _Scratch:
    """GroupNorm scratch sized for the largest call of one encode / decode."""

    def __init__(self):
        self.bufs = {}

    def get(self, M, rows_per_group, device, groups=32):
        """One buffer per scratch slot (``ops.workspace_slot``): the clip pipeline encodes the next clip on one stream while it
        decodes the current one on another."""
        need = ops.gn_scratch_floats(M, rows_per_group, groups)
        buf = self.bufs.get(ops.WS_SLOT)
        if buf is None or buf.numel() < need or buf.device != device:
            buf = self.bufs[ops.WS_SLOT] = torch.empty(need, dtype=torch.float32, device=device)
        return buf


class VAEResnetBlock(nn.Module):
    """ResnetBlock2D without time embedding (eps 1e-6): GN+SiLU -> conv -> GN+SiLU -> conv (+ shortcut, fused)."""

    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = Conv2d(cin, cout, 3, padding=1)
        self.norm2 = GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = Conv2d(cin, cout, 1) if cin != cout else None

    def run(self, sc: _Scratch, x, H, W):
        HW, g = H * W, self.norm1.num_groups
        h = ops.groupnorm(x, self.norm1.weight, self.norm1.bias, sc.get(x.shape[0], HW, x.device, g), HW, groups=g,
                          eps=self.norm1.eps, silu=True)
        h = self.conv1.tokens(h, H, W)
        h = ops.groupnorm(h, self.norm2.weight, self.norm2.bias, sc.get(h.shape[0], HW, x.device, g), HW, groups=g,
                          eps=self.norm2.eps, silu=True)
        res = x if self.conv_shortcut is None else self.conv_shortcut.tokens(x, H, W)
        return self.conv2.tokens(h, H, W, residual=res)


class VAEAttention(nn.Module):
    """One head of width C over the H*W tokens of each image, residual connection."""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = Linear(c, c), Linear(c, c), Linear(c, c)
        self.to_out = nn.ModuleList([Linear(c, c), nn.Identity()])
        self.scale = c ** -0.5

    def run(self, sc: _Scratch, x, H, W):
        HW, g = H * W, self.group_norm.num_groups
        n = x.shape[0] // HW
        h = ops.groupnorm(x, self.group_norm.weight, self.group_norm.bias, sc.get(x.shape[0], HW, x.device, g), HW, groups=g,
                          eps=self.group_norm.eps)
        q = ops.gemm(h, self.to_q.weight, bias=self.to_q.bias)
        k = ops.gemm(h, self.to_k.weight, bias=self.to_k.bias)
        v = ops.gemm(h, self.to_v.weight, bias=self.to_v.bias)
        o = torch.empty_like(q)
        for i in range(n):  # one image at a time: logits [HW, HW] fp32 (64 MiB at 64x64 tokens)
            r = slice(i * HW, (i + 1) * HW)
            logits = ops.gemm(q[r], k[r], act=ACT_F32OUT)
            p = ops.softmax_rows(logits, self.scale)
            ops.gemm(p, v[r].t().contiguous(), out=o[r])
        return ops.gemm(o, self.to_out[0].weight, bias=self.to_out[0].bias, residual=x)


class VAEMidBlock(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([VAEResnetBlock(c, c, groups), VAEResnetBlock(c, c, groups)])
        self.attentions = nn.ModuleList([VAEAttention(c, groups)])

    def run(self, sc, x, H, W):
        x = self.resnets[0].run(sc, x, H, W)
        x = self.attentions[0].run(sc, x, H, W)
        return self.resnets[1].run(sc, x, H, W)


class _Sampler(nn.Module):
    def __init__(self, c, stride):
        super().__init__()
        self.conv = Conv2d(c, c, 3, stride=stride, padding=1)


class VAEDownBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([VAEResnetBlock(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([_Sampler(cout, 2)]) if down else None

    def run(self, sc, x, H, W):
        for r in self.resnets:
            x = r.run(sc, x, H, W)
        if self.downsamplers is not None:
            x = self.downsamplers[0].conv.tokens(x, H, W, asym=True)  # F.pad (0,1,0,1) + stride-2 conv
            H, W = H // 2, W // 2
        return x, H, W


class VAEUpBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([VAEResnetBlock(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.upsamplers = nn.ModuleList([_Sampler(cout, 1)]) if up else None

    def run(self, sc, x, H, W):
        for r in self.resnets:
            x = r.run(sc, x, H, W)
        if self.upsamplers is not None:
            x = self.upsamplers[0].conv.tokens(x, H, W, up=True)  # nearest x2 folded into the conv gather
            H, W = 2 * H, 2 * W
        return x, H, W


class VAEEncoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = Conv2d(cfg.in_channels, boc[0], 3, padding=1, pad_cin_to=PAD_CIN)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, c in enumerate(boc):
            cin, out = out, c
            self.down_blocks.append(VAEDownBlock(cin, out, cfg.layers_per_block, g, i != len(boc) - 1))
        self.mid_block = VAEMidBlock(boc[-1], g)
        self.conv_norm_out = GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = Conv2d(boc[-1], 2 * cfg.latent_channels, 3, padding=1)


class VAEDecoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        rev = list(reversed(boc))
        self.conv_in = Conv2d(cfg.latent_channels, rev[0], 3, padding=1, pad_cin_to=PAD_CIN)
        self.mid_block = VAEMidBlock(rev[0], g)
        self.up_blocks = nn.ModuleList()
        out = rev[0]
        for i, c in enumerate(rev):
            cin, out = out, c
            self.up_blocks.append(VAEUpBlock(cin, out, cfg.layers_per_block + 1, g, i != len(boc) - 1))
        self.conv_norm_out = GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = Conv2d(boc[0], cfg.out_channels, 3, padding=1)


class AutoencoderKL(nn.Module):
    def __init__(self, cfg: Optional[VAEConfig] = None):
        super().__init__()
        self.cfg = cfg or VAEConfig()
        self.config = self.cfg
        self.encoder, self.decoder = VAEEncoder(self.cfg), VAEDecoder(self.cfg)
        self.quant_conv = Conv2d(2 * self.cfg.latent_channels, 2 * self.cfg.latent_channels, 1)
        self.post_quant_conv = Conv2d(self.cfg.latent_channels, self.cfg.latent_channels, 1)
        self._packed = False
        self._sc = _Scratch()

    def load_state_dict(self, sd, strict=True, **kw):
        r = super().load_state_dict({k: v.to(torch.float16) for k, v in sd.items()}, strict=strict, **kw)
        self._packed = False
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._packed = False
        return r

    def pack(self):
        for m in self.modules():
            if m is not self and hasattr(m, "pack"):
                m.pack()
        self._packed = True

    # -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_moments(self, x: torch.Tensor):
        """[n,3,H,W] in [-1,1] (any float dtype, on the module's device) -> (mean, logvar) fp32 [n,4,H/8,W/8]."""
        if not self._packed:
            self.pack()
        n, c, H, W = x.shape
        dev = self.quant_conv.weight.device
        tok = torch.zeros((n * H * W, PAD_CIN), dtype=torch.float16, device=dev)
        ops.ncfhw_to_tokens(x.to(device=dev, dtype=torch.float16).permute(1, 0, 2, 3)[None].contiguous(), tok, col0=0)
        enc, sc = self.encoder, self._sc
        h = enc.conv_in.tokens(tok, H, W)
        for b in enc.down_blocks:
            h, H, W = b.run(sc, h, H, W)
        h = enc.mid_block.run(sc, h, H, W)
        g = enc.conv_norm_out
        h = ops.groupnorm(h, g.weight, g.bias, sc.get(h.shape[0], H * W, dev, g.num_groups), H * W, groups=g.num_groups,
                          eps=g.eps, silu=True)
        m = enc.conv_out.tokens(h, H, W)                      # [n h w, 8]
        m = self.quant_conv.tokens(m, H, W)                   # 1x1 conv
        L = self.cfg.latent_channels
        m = m.float().view(n, H, W, 2 * L).permute(0, 3, 1, 2)
        return m[:, :L].contiguous(), m[:, L:].clamp(-30.0, 20.0).contiguous()

    @torch.no_grad()
    def decode(self, z: torch.Tensor):
        """[n,4,h,w] (already divided by scaling_factor) -> fp32 [n,3,8h,8w]."""
        if not self._packed:
            self.pack()
        n, L, H, W = z.shape
        dev = self.quant_conv.weight.device
        z16 = torch.zeros((n * H * W, 8), dtype=torch.float16, device=dev)
        ops.ncfhw_to_tokens(z.to(device=dev, dtype=torch.float16).permute(1, 0, 2, 3)[None].contiguous(), z16, col0=0)
        zq = self.post_quant_conv.tokens(z16[:, :L].contiguous(), H, W)  # 1x1 conv on 4 channels (reference-grade kernel)
        tok = torch.zeros((n * H * W, PAD_CIN), dtype=torch.float16, device=dev)
        ops.copy_cols(zq, 0, tok, 0, L)
        dec, sc = self.decoder, self._sc
        h = dec.conv_in.tokens(tok, H, W)
        h = dec.mid_block.run(sc, h, H, W)
        for b in dec.up_blocks:
            h, H, W = b.run(sc, h, H, W)
        g = dec.conv_norm_out
        h = ops.groupnorm(h, g.weight, g.bias, sc.get(h.shape[0], H * W, dev, g.num_groups), H * W, groups=g.num_groups,
                          eps=g.eps, silu=True)
        out = torch.empty((n * H * W, 8), dtype=torch.float16, device=dev)
        dec.conv_out.tokens(h, H, W, out=out)
        img = ops.tokens_to_ncfhw(out, 1, self.cfg.out_channels, n, H, W)  # [1,3,n,H,W]
        return img[0].permute(1, 0, 2, 3).float().contiguous()


def init_random_weights_(vae: AutoencoderKL, seed: int = 0):
    """Seeded fan-in-scaled weights (offline stand-in for the checkpoint; real weights: ``load_state_dict``)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in vae.named_parameters():
            if p.dim() >= 2:
                w = torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5
            elif name.endswith("weight"):
                w = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            else:
                w = 0.05 * torch.randn(p.shape, generator=g)
            p.copy_(w.to(p.dtype))
    vae._packed = False
    return vae
