"""Pure-PyTorch restatement of diffusers-0.26.3 ``I2VGenXLUNet`` (TEST ORACLE, not product).

The reference imports the model from a third-party package that is absent here
(``/root/reference/i2vgen-xl/pipelines/pipeline_i2vgen_xl.py:29``; pinned
``diffusers==0.26.3`` at ``i2vgen-xl/environment.yml:15``) and calls it at
``pipeline_i2vgen_xl.py:845,1146,1395``.  This file restates the published
architecture with the same module tree / state-dict keys so
that (a) real ``ali-vilab/i2vgen-xl`` weights would load unchanged and (b) the
reference's own ``i2vgen-xl/pnp_utils.py`` attaches to it unmodified
(attribute paths used there: ``pnp_utils.py:20-27,130,239,344``).

In-repo evidence each block follows:
  * ResnetBlock2D body ............ ``i2vgen-xl/pnp_utils.py:46-126``
  * AttnProcessor2_0 body .......... ``i2vgen-xl/pnp_utils.py:151-228``
  * BasicTransformerBlock order .... ``consisti2v/consisti2v/models/videoldm_transformer_blocks.py:461-564``
  * Transformer2D wrapper .......... ``consisti2v/.../videoldm_transformer_blocks.py:222-280``
  * up-block skip arithmetic ....... ``consisti2v/consisti2v/models/videoldm_unet_blocks.py:599-606,721-745``
  * temporal (3,1,1) conv .......... ``consisti2v/.../videoldm_unet_blocks.py:316-328``

Everything is parametrised by ``UNetConfig`` so tests can run a narrow ("mini")
network on CPU in seconds; ``UNetConfig.i2vgen_xl()`` is the real 1.42 B model.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    cross_attention_dim: int = 1024
    attention_head_dim: int = 64          # hub config's "num_attention_heads=64" is used as head dim
    transformer_in_heads: int = 8
    sample_size: int = 32
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock3D",) * 3 + ("DownBlock3D",)
    up_block_types: Tuple[str, ...] = ("UpBlock3D",) + ("CrossAttnUpBlock3D",) * 3

    @staticmethod
    def i2vgen_xl() -> "UNetConfig":
        return UNetConfig()

    @staticmethod
    def mini() -> "UNetConfig":
        """Same graph, narrow channels (head dim stays 64, GroupNorm stays 32 groups)."""
        return UNetConfig(block_out_channels=(64, 128, 256, 256), cross_attention_dim=128,
                          transformer_in_heads=2, sample_size=8)

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4


# --------------------------------------------------------------------------- attention
class AttnProcessor2_0:
    """diffusers ``AttnProcessor2_0`` == ``i2vgen-xl/pnp_utils.py:151-228`` minus the injection block."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0):
        batch_size = hidden_states.shape[0]
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        head_dim = key.shape[-1] // attn.heads
        query = query.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        key = key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        value = value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False)
        hidden_states = hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim)
        hidden_states = hidden_states.to(query.dtype)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        return hidden_states


class Attention(nn.Module):
    """diffusers ``Attention`` with the attribute surface the reference hooks read."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        ctx = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_k = nn.Linear(ctx, inner, bias=False)
        self.to_v = nn.Linear(ctx, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])
        # attributes the hooks touch
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = AttnProcessor2_0()

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        return attention_mask

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class GELUProj(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)

    def forward(self, x):
        return F.gelu(self.proj(x))


class FeedForward(nn.Module):
    def __init__(self, dim, inner_dim=None, activation_fn="geglu"):
        super().__init__()
        inner_dim = inner_dim or dim * 4
        act = GEGLU(dim, inner_dim) if activation_fn == "geglu" else GELUProj(dim, inner_dim)
        self.net = nn.ModuleList([act, nn.Dropout(0.0), nn.Linear(inner_dim, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim=None, double_self_attention=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, None if double_self_attention else cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, encoder_hidden_states=None):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states=encoder_hidden_states) + x
        x = self.ff(self.norm3(x)) + x
        return x


class Transformer2DModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, encoder_hidden_states):
        n, c, h, w = x.shape
        res = x
        x = self.norm(x)
        x = x.permute(0, 2, 3, 1).reshape(n, h * w, c)
        x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states)
        x = self.proj_out(x)
        x = x.reshape(n, h, w, c).permute(0, 3, 1, 2).contiguous()
        return x + res


class TransformerTemporalModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, groups):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, None, double_self_attention=True)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, num_frames):
        bf, c, h, w = x.shape
        b = bf // num_frames
        res = x
        x = x[None, :].reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        x = self.norm(x)
        x = x.permute(0, 3, 4, 2, 1).reshape(b * h * w, num_frames, c)
        x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, None)
        x = self.proj_out(x)
        x = x[None, None, :].reshape(b, h, w, num_frames, c).permute(0, 3, 4, 1, 2).contiguous()
        x = x.reshape(bf, c, h, w)
        return x + res


# --------------------------------------------------------------------------- conv blocks
class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x, output_size=None):
        """[3P] diffusers ``Upsample2D.forward(hidden_states, output_size)``: nearest x2, or -- when the UNet forwards the size of the
        skip connection it is about to meet -- nearest to exactly that size (in-tree copies of the same rule:
        ``seine/models/resnet.py:44-64``)."""
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=tuple(output_size), mode="nearest")
        return self.conv(x)


class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x, scale=1.0):
        return self.conv(x)


class ResnetBlock2D(nn.Module):
    """Body == ``i2vgen-xl/pnp_utils.py:46-126`` (time_embedding_norm="default")."""

    def __init__(self, in_channels, out_channels, temb_channels, groups, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.nonlinearity = nn.SiLU()
        self.upsample = self.downsample = None
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.skip_time_act = False
        self.time_embedding_norm = "default"
        self.output_scale_factor = 1.0

    def forward(self, input_tensor, temb, scale=1.0):
        h = self.nonlinearity(self.norm1(input_tensor))
        h = self.conv1(h)
        t = self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = h + t
        h = self.nonlinearity(self.norm2(h))
        h = self.conv2(self.dropout(h))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


class TemporalConvLayer(nn.Module):
    def __init__(self, dim, groups):
        super().__init__()
        self.conv1 = nn.Sequential(nn.GroupNorm(groups, dim), nn.SiLU(),
                                   nn.Conv3d(dim, dim, (3, 1, 1), padding=(1, 0, 0)))
        for name in ("conv2", "conv3", "conv4"):
            setattr(self, name, nn.Sequential(nn.GroupNorm(groups, dim), nn.SiLU(), nn.Dropout(0.1),
                                              nn.Conv3d(dim, dim, (3, 1, 1), padding=(1, 0, 0))))

    def forward(self, x, num_frames):
        x = x[None, :].reshape((-1, num_frames) + x.shape[1:]).permute(0, 2, 1, 3, 4)
        identity = x
        x = self.conv4(self.conv3(self.conv2(self.conv1(x))))
        x = identity + x
        return x.permute(0, 2, 1, 3, 4).reshape((x.shape[0] * x.shape[2], -1) + x.shape[3:])


class DownBlock3D(nn.Module):
    def __init__(self, cfg: UNetConfig, cin, cout, cross_attn: bool, add_downsample: bool):
        super().__init__()
        g, temb = cfg.norm_num_groups, cfg.time_embed_dim
        self.has_cross_attention = cross_attn
        self.resnets = nn.ModuleList()
        self.temp_convs = nn.ModuleList()
        if cross_attn:
            self.attentions = nn.ModuleList()
            self.temp_attentions = nn.ModuleList()
        for i in range(cfg.layers_per_block):
            self.resnets.append(ResnetBlock2D(cin if i == 0 else cout, cout, temb, g))
            self.temp_convs.append(TemporalConvLayer(cout, g))
            if cross_attn:
                heads = cout // cfg.attention_head_dim
                self.attentions.append(Transformer2DModel(heads, cfg.attention_head_dim, cout, cfg.cross_attention_dim, g))
                self.temp_attentions.append(TransformerTemporalModel(heads, cfg.attention_head_dim, cout, g))
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None

    def forward(self, x, temb, ctx, num_frames):
        outs = []
        for i, (resnet, tconv) in enumerate(zip(self.resnets, self.temp_convs)):
            x = resnet(x, temb)
            x = tconv(x, num_frames)
            if self.has_cross_attention:
                x = self.attentions[i](x, ctx)
                x = self.temp_attentions[i](x, num_frames)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock3D(nn.Module):
    def __init__(self, cfg: UNetConfig, c):
        super().__init__()
        g, temb = cfg.norm_num_groups, cfg.time_embed_dim
        heads = c // cfg.attention_head_dim
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, g), ResnetBlock2D(c, c, temb, g)])
        self.temp_convs = nn.ModuleList([TemporalConvLayer(c, g), TemporalConvLayer(c, g)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, cfg.attention_head_dim, c, cfg.cross_attention_dim, g)])
        self.temp_attentions = nn.ModuleList([TransformerTemporalModel(heads, cfg.attention_head_dim, c, g)])

    def forward(self, x, temb, ctx, num_frames):
        x = self.resnets[0](x, temb)
        x = self.temp_convs[0](x, num_frames)
        x = self.attentions[0](x, ctx)
        x = self.temp_attentions[0](x, num_frames)
        x = self.resnets[1](x, temb)
        x = self.temp_convs[1](x, num_frames)
        return x


class UpBlock3D(nn.Module):
    def __init__(self, cfg: UNetConfig, cin, cout, prev_out, cross_attn: bool, add_upsample: bool):
        super().__init__()
        g, temb = cfg.norm_num_groups, cfg.time_embed_dim
        n = cfg.layers_per_block + 1
        self.has_cross_attention = cross_attn
        self.resnets = nn.ModuleList()
        self.temp_convs = nn.ModuleList()
        if cross_attn:
            self.attentions = nn.ModuleList()
            self.temp_attentions = nn.ModuleList()
        for i in range(n):
            # consisti2v/.../videoldm_unet_blocks.py:599-606
            res_skip = cin if i == n - 1 else cout
            res_in = prev_out if i == 0 else cout
            self.resnets.append(ResnetBlock2D(res_in + res_skip, cout, temb, g))
            self.temp_convs.append(TemporalConvLayer(cout, g))
            if cross_attn:
                heads = cout // cfg.attention_head_dim
                self.attentions.append(Transformer2DModel(heads, cfg.attention_head_dim, cout, cfg.cross_attention_dim, g))
                self.temp_attentions.append(TransformerTemporalModel(heads, cfg.attention_head_dim, cout, g))
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None

    def forward(self, x, skips: List[torch.Tensor], temb, ctx, num_frames, upsample_size=None):
        for i, (resnet, tconv) in enumerate(zip(self.resnets, self.temp_convs)):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet(x, temb)
            x = tconv(x, num_frames)
            if self.has_cross_attention:
                x = self.attentions[i](x, ctx)
                x = self.temp_attentions[i](x, num_frames)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x, upsample_size)
        return x


# --------------------------------------------------------------------------- top level
def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class I2VGenXLTransformerTemporalEncoder(nn.Module):
    def __init__(self, dim, heads, dim_head, ff_inner):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.ff = FeedForward(dim, ff_inner, activation_fn="gelu")

    def forward(self, x):
        x = self.attn1(self.norm1(x)) + x
        x = self.ff(x) + x
        return x


class _Cfg:
    """Mimics ``unet.config`` attribute access used at pipeline_i2vgen_xl.py:743,820."""

    def __init__(self, cfg: UNetConfig):
        self.in_channels = cfg.in_channels
        self.sample_size = cfg.sample_size
        self.cross_attention_dim = cfg.cross_attention_dim


class I2VGenXLUNetOracle(nn.Module):
    def __init__(self, cfg: Optional[UNetConfig] = None):
        super().__init__()
        cfg = cfg or UNetConfig.i2vgen_xl()
        self.cfg = cfg
        self.config = _Cfg(cfg)
        boc = cfg.block_out_channels
        g = cfg.norm_num_groups
        ic = cfg.in_channels
        ted = cfg.time_embed_dim

        self.conv_in = nn.Conv2d(ic + ic, boc[0], 3, padding=1)
        self.transformer_in = TransformerTemporalModel(cfg.transformer_in_heads, cfg.attention_head_dim, boc[0], g)
        self.image_latents_proj_in = nn.Sequential(
            nn.Conv2d(4, ic * 4, 3, padding=1), nn.SiLU(),
            nn.Conv2d(ic * 4, ic * 4, 3, padding=1), nn.SiLU(),
            nn.Conv2d(ic * 4, ic, 3, padding=1))
        self.image_latents_temporal_encoder = I2VGenXLTransformerTemporalEncoder(ic, 2, ic, ic * 4)
        self.image_latents_context_embedding = nn.Sequential(
            nn.Conv2d(4, ic * 8, 3, padding=1), nn.SiLU(), nn.AdaptiveAvgPool2d((32, 32)),
            nn.Conv2d(ic * 8, ic * 16, 3, stride=2, padding=1), nn.SiLU(),
            nn.Conv2d(ic * 16, cfg.cross_attention_dim, 3, stride=2, padding=1))
        self.time_embedding = nn.ModuleDict(dict(linear_1=nn.Linear(boc[0], ted), linear_2=nn.Linear(ted, ted)))
        self.context_embedding = nn.Sequential(nn.Linear(cfg.cross_attention_dim, ted), nn.SiLU(),
                                               nn.Linear(ted, cfg.cross_attention_dim * ic))
        self.fps_embedding = nn.Sequential(nn.Linear(boc[0], ted), nn.SiLU(), nn.Linear(ted, ted))

        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, typ in enumerate(cfg.down_block_types):
            cin, out = out, boc[i]
            last = i == len(boc) - 1
            self.down_blocks.append(DownBlock3D(cfg, cin, out, typ.startswith("CrossAttn"), not last))
        self.mid_block = MidBlock3D(cfg, boc[-1])
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out = rev[0]
        for i, typ in enumerate(cfg.up_block_types):
            last = i == len(boc) - 1
            prev_out, out = out, rev[i]
            cin = rev[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(UpBlock3D(cfg, cin, out, prev_out, typ.startswith("CrossAttn"), not last))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def forward(self, sample, timestep, fps=None, image_latents=None, image_embeddings=None,
                encoder_hidden_states=None, cross_attention_kwargs=None, return_dict=False):
        cfg = self.cfg
        B, C, Fr, H, W = sample.shape
        dt = self.dtype
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.long, device=sample.device)
        timestep = timestep.reshape(-1).expand(B)
        # 1-3: time + fps embeddings
        t_emb = timestep_embedding(timestep, cfg.block_out_channels[0]).to(dt)
        t_emb = self.time_embedding["linear_2"](F.silu(self.time_embedding["linear_1"](t_emb)))
        fps = fps.reshape(-1).expand(B)
        fps_emb = self.fps_embedding(timestep_embedding(fps, cfg.block_out_channels[0]).to(dt))
        emb = (t_emb + fps_emb).repeat_interleave(Fr, dim=0)
        # 4: context = [text | first-frame-latent tokens | CLIP-image tokens]
        il0 = image_latents[:, :, :1].permute(0, 2, 1, 3, 4).reshape(B, C, H, W)
        il_ctx = self.image_latents_context_embedding(il0)
        _b, _c, _h, _w = il_ctx.shape
        il_ctx = il_ctx.permute(0, 2, 3, 1).reshape(_b, _h * _w, _c)
        img_emb = self.context_embedding(image_embeddings).view(-1, cfg.in_channels, cfg.cross_attention_dim)
        ctx = torch.cat([encoder_hidden_states, il_ctx, img_emb], dim=1).repeat_interleave(Fr, dim=0)
        # image latents branch
        il = image_latents.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W)
        il = self.image_latents_proj_in(il)
        il = il[None, :].reshape(B, Fr, C, H, W).permute(0, 3, 4, 1, 2).reshape(B * H * W, Fr, C)
        il = self.image_latents_temporal_encoder(il)
        il = il.reshape(B, H, W, Fr, C).permute(0, 4, 3, 1, 2)
        # 5: pre-process
        x = torch.cat([sample, il], dim=1)
        x = x.permute(0, 2, 1, 3, 4).reshape(B * Fr, -1, H, W)
        x = self.conv_in(x)
        x = self.transformer_in(x, Fr)
        # 6: down
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, ctx, Fr)
            skips.extend(outs)
        # 7: mid
        x = self.mid_block(x, emb, ctx, Fr)
        # 8: up.  [3P] a latent size that is not a multiple of 2 ** (number of upsamplers) = 8 does not come back from three
        # ceil-halvings by doubling: the UNet then hands every up block the spatial size of the skip connection the NEXT block pops
        # first (``forward_upsample_size`` / ``upsample_size``; in-tree copies of the rule: ``seine/models/unet.py:393-401,485-500``,
        # ``consisti2v/consisti2v/models/videoldm_unet.py:726-734,990-1010``)
        forward_size = any(s % (2 ** (len(self.up_blocks) - 1)) != 0 for s in (H, W))
        for bi, blk in enumerate(self.up_blocks):
            n_pop = len(blk.resnets)
            size = tuple(skips[-n_pop - 1].shape[2:]) if (forward_size and bi != len(self.up_blocks) - 1) else None
            x = blk(x, skips, emb, ctx, Fr, upsample_size=size)
        # 9: post
        x = self.conv_out(self.conv_act(self.conv_norm_out(x)))
        x = x[None, :].reshape((-1, Fr) + x.shape[1:]).permute(0, 2, 1, 3, 4)
        return (x,)


# --------------------------------------------------------------------------- weights
def random_state_dict(cfg: UNetConfig, seed: int = 0, dtype=torch.float16) -> "dict[str, torch.Tensor]":
    """Seeded random weights for the exact architecture, rounded to ``dtype``.

    No pretrained weights exist offline.  Init: N(0, 1/fan_in) matrices,
    N(0, 0.02) biases, N(1, 0.05) norm gains -- keeps activations O(1) through ~100
    residual layers so fp16 ranges look like a trained network's.  (The real model
    zero-initialises the last temporal conv; random here so the path is exercised.)
    """
    with torch.device("meta"):
        m = I2VGenXLUNetOracle(cfg)
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in m.state_dict().items():
        shp = tuple(v.shape)
        if k.endswith(".bias"):
            t = torch.randn(shp, generator=gen) * 0.02
        elif len(shp) == 1:
            t = 1.0 + torch.randn(shp, generator=gen) * 0.05
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = torch.randn(shp, generator=gen) / math.sqrt(fan_in)
        sd[k] = t.to(dtype)
    return sd


def build_oracle(cfg: UNetConfig, state_dict, dtype=torch.float32, device="cpu") -> I2VGenXLUNetOracle:
    with torch.device("meta"):
        m = I2VGenXLUNetOracle(cfg)
    m = m.to_empty(device=device)
    m.load_state_dict({k: v.to(device=device, dtype=dtype) for k, v in state_dict.items()}, strict=True)
    return m.to(dtype).eval()


def build_random_oracle(cfg: UNetConfig, seed: int = 0) -> I2VGenXLUNetOracle:
    """fp32 oracle with in-place random init (same law as ``random_state_dict``; values differ).  Fast path for the
    CPU-baseline timing in bench.py, where only the arithmetic volume matters."""
    with torch.device("meta"):
        m = I2VGenXLUNetOracle(cfg)
    m = m.to_empty(device="cpu")
    torch.manual_seed(seed)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith(".bias"):
                p.normal_(0.0, 0.02)
            elif p.dim() == 1:
                p.normal_(1.0, 0.05)
            else:
                fan_in = 1
                for d in p.shape[1:]:
                    fan_in *= d
                p.normal_(0.0, 1.0 / math.sqrt(fan_in))
    return m.eval()


def param_count(cfg: UNetConfig) -> int:
    with torch.device("meta"):
        m = I2VGenXLUNetOracle(cfg)
    return sum(p.numel() for p in m.parameters())
