"""CLIP text / vision towers of the ``ali-vilab/i2vgen-xl`` checkpoint on the HIP kernels -- SURVEY.md 8(f) F1.

Replaces ``self.text_encoder`` in ``encode_prompt`` (``i2vgen-xl/pipelines/pipeline_i2vgen_xl.py:224-409``; ``clip_skip``
semantics ``:312-324``: hidden state ``-(clip_skip + 1)`` followed by ``final_layer_norm``) and ``self.image_encoder`` in
``_encode_image`` (``:411-441``: ``image_embeds`` of ``CLIPVisionModelWithProjection``).  Once per clip, outside the loops.

State-dict keys are those of ``transformers``' ``CLIPTextModel`` / ``CLIPVisionModelWithProjection``
(``text_model.encoder.layers.0.self_attn.q_proj.weight`` ...), so ``text_encoder/model.safetensors`` and
``image_encoder/model.safetensors`` of the checkpoint load as they are.  Layout and kernels are the UNet's: token matrices
``[(b s), C]`` fp16; LayerNorm kernel; one fused-QKV GEMM, the generic attention kernel (causal mask for the text tower,
head_dim 80 for ViT-H/14), out-proj GEMM with the residual fused; ``fc1`` with the erf-GELU epilogue, ``fc2`` with the residual
fused.  The token / position embedding lookups and the patch unfold are index / reshape operations on device tensors.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import ops


class CLIPTowerConfig:
    def __init__(self, hidden_size, num_hidden_layers, num_attention_heads, intermediate_size, layer_norm_eps=1e-5,
                 hidden_act="gelu", vocab_size=None, max_position_embeddings=None, image_size=None, patch_size=None,
                 num_channels=3, projection_dim=None, **_ignored):
        self.hidden_size, self.num_hidden_layers, self.num_attention_heads = hidden_size, num_hidden_layers, num_attention_heads
        self.intermediate_size, self.layer_norm_eps, self.hidden_act = intermediate_size, layer_norm_eps, hidden_act
        self.vocab_size, self.max_position_embeddings = vocab_size, max_position_embeddings
        self.image_size, self.patch_size, self.num_channels, self.projection_dim = image_size, patch_size, num_channels, projection_dim
        if hidden_act != "gelu":
            # the i2vgen-xl towers (OpenCLIP ViT-H/14) use erf-GELU, which is the GEMM kernel's ACT_GELU epilogue;
            # quick_gelu (OpenAI CLIP) has no kernel here
            raise NotImplementedError(f"CLIP hidden_act={hidden_act!r}: only 'gelu' (OpenCLIP ViT-H/14) runs on the HIP kernels")
        if hidden_size % num_attention_heads or hidden_size // num_attention_heads > 128:
            raise ValueError("CLIP head_dim must divide hidden_size and be <= 128")

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @classmethod
    def from_hf(cls, cfg: dict, tower: Optional[str] = None):
        """From a transformers ``config.json`` dict (a tower config, or a full CLIPConfig with text_config / vision_config)."""
        if tower is not None and f"{tower}_config" in cfg:
            cfg = cfg[f"{tower}_config"]
        return cls(**cfg)


class _Tower:
    """Shared encoder stack.  ``P``: fp16 device tensors under the transformers key names (+ packed fused-QKV copies)."""

    def __init__(self, cfg: CLIPTowerConfig, state_dict: Dict[str, torch.Tensor], prefix: str):
        self.cfg, self.prefix = cfg, prefix
        # checkpoints written by transformers 4.x nest the tower under ``text_model.`` / ``vision_model.``; transformers >= 5
        # flattens a bare CLIPTextModel's keys (``embeddings...``, ``encoder...``): accept both
        def canon(k):
            return k if k.startswith(prefix + ".") or k.split(".")[0].endswith("_projection") else f"{prefix}.{k}"
        self.P = {canon(k): v.detach().to(torch.float16) for k, v in state_dict.items()
                  if torch.is_tensor(v) and v.is_floating_point()}
        self._packed = False
        self.device = torch.device("cpu")

    def to(self, device):
        self.P = {k: v.to(device).contiguous() for k, v in self.P.items()}
        self.device = torch.device(device)
        if not self._packed:
            for i in range(self.cfg.num_hidden_layers):
                a = f"{self.prefix}.encoder.layers.{i}.self_attn."
                self.P[a + "qkv.weight"] = torch.cat([self.P[a + f"{n}_proj.weight"] for n in "qkv"]).contiguous()
                self.P[a + "qkv.bias"] = torch.cat([self.P[a + f"{n}_proj.bias"] for n in "qkv"]).contiguous()
            self._packed = True
        return self

    def ensure(self, device):
        """Move / pack once for ``device`` (no-op afterwards)."""
        if not self._packed or self.device != torch.device(device):
            self.to(device)
        return self

    def layer(self, x: torch.Tensor, i: int, B: int, S: int, causal: bool) -> torch.Tensor:
        """One pre-LN CLIP encoder layer on the token matrix ``x`` [(B S), H] (transformers ``CLIPEncoderLayer``)."""
        cfg, P = self.cfg, self.P
        L = f"{self.prefix}.encoder.layers.{i}."
        H, h, d = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim
        y = ops.layernorm(x, P[L + "layer_norm1.weight"], P[L + "layer_norm1.bias"], cfg.layer_norm_eps)
        qkv = ops.gemm(y, P[L + "self_attn.qkv.weight"], bias=P[L + "self_attn.qkv.bias"])
        o = torch.empty(B * S, H, dtype=torch.float16, device=x.device)
        ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], o, batch=B, heads=h, Sq=S, Sk=S, inner=1,
                      q_strides=(S, 0, 1), kv_strides=(S, 0, 1), scale=d ** -0.5, head_dim=d, causal=causal)
        x = ops.gemm(o, P[L + "self_attn.out_proj.weight"], bias=P[L + "self_attn.out_proj.bias"], residual=x)
        y = ops.layernorm(x, P[L + "layer_norm2.weight"], P[L + "layer_norm2.bias"], cfg.layer_norm_eps)
        y = ops.gemm(y, P[L + "mlp.fc1.weight"], bias=P[L + "mlp.fc1.bias"], act=ops.ACT_GELU)
        return ops.gemm(y, P[L + "mlp.fc2.weight"], bias=P[L + "mlp.fc2.bias"], residual=x)


class CLIPTextTower(_Tower):
    def __init__(self, cfg: CLIPTowerConfig, state_dict):
        super().__init__(cfg, state_dict, "text_model")

    @torch.no_grad()
    def hidden_states(self, input_ids: torch.Tensor) -> List[torch.Tensor]:
        """``CLIPTextModel(..., output_hidden_states=True).hidden_states``: embeddings output, then every layer's output,
        each [B, S, H] (causal self-attention; the final LayerNorm is NOT applied)."""
        B, S = input_ids.shape
        P, cfg = self.P, self.cfg
        ids = input_ids.to(self.device)
        tok = P["text_model.embeddings.token_embedding.weight"][ids.reshape(-1)]
        pos = P["text_model.embeddings.position_embedding.weight"][:S].repeat(B, 1)
        x = ops.add(tok.contiguous(), pos.contiguous())
        out = [x.view(B, S, -1)]
        for i in range(cfg.num_hidden_layers):
            x = self.layer(x, i, B, S, causal=True)
            out.append(x.view(B, S, -1))
        return out

    @torch.no_grad()
    def final_layer_norm(self, h: torch.Tensor) -> torch.Tensor:
        B, S, H = h.shape
        P = self.P
        return ops.layernorm(h.reshape(B * S, H).contiguous(), P["text_model.final_layer_norm.weight"],
                             P["text_model.final_layer_norm.bias"], self.cfg.layer_norm_eps).view(B, S, H)

    @torch.no_grad()
    def encode_ids(self, input_ids: torch.Tensor, clip_skip: Optional[int] = None) -> torch.Tensor:
        """``pipeline_i2vgen_xl.py:305-324``: last_hidden_state, or hidden state -(clip_skip + 1) + final_layer_norm."""
        hs = self.hidden_states(input_ids)
        return self.final_layer_norm(hs[-1] if clip_skip is None else hs[-(clip_skip + 1)])


class CLIPVisionTower(_Tower):
    def __init__(self, cfg: CLIPTowerConfig, state_dict):
        super().__init__(cfg, state_dict, "vision_model")

    def to(self, device):
        super().to(device)
        w = self.P["vision_model.embeddings.patch_embedding.weight"]
        self.P["vision_model.embeddings.patch_embedding.packed"] = w.reshape(w.shape[0], -1).contiguous()  # [H, 3 P P]
        return self

    @torch.no_grad()
    def image_embeds(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """``CLIPVisionModelWithProjection(pixel_values).image_embeds``: [B, projection_dim]."""
        cfg, P = self.cfg, self.P
        B, C, Hh, Ww = pixel_values.shape
        p = cfg.patch_size
        gh, gw = Hh // p, Ww // p
        x = pixel_values.to(self.device, torch.float16)
        # non-overlapping p x p patches -> rows [(b gy gx), (c py px)]: the stride-p convolution as one GEMM
        patches = x.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, C * p * p).contiguous()
        emb = ops.gemm(patches, P["vision_model.embeddings.patch_embedding.packed"])          # [(b n), H], no bias
        H = cfg.hidden_size
        S = gh * gw + 1
        tokens = torch.empty(B, S, H, dtype=torch.float16, device=x.device)
        tokens[:, 0] = P["vision_model.embeddings.class_embedding"]
        tokens[:, 1:] = emb.view(B, gh * gw, H)
        pos = P["vision_model.embeddings.position_embedding.weight"][:S].repeat(B, 1)
        xt = ops.add(tokens.view(B * S, H), pos.contiguous())
        xt = ops.layernorm(xt, P["vision_model.pre_layrnorm.weight"], P["vision_model.pre_layrnorm.bias"], cfg.layer_norm_eps)
        for i in range(cfg.num_hidden_layers):
            xt = self.layer(xt, i, B, S, causal=False)
        pooled = xt.view(B, S, H)[:, 0].contiguous()
        pooled = ops.layernorm(pooled, P["vision_model.post_layernorm.weight"], P["vision_model.post_layernorm.bias"],
                               cfg.layer_norm_eps)
        return ops.gemm(pooled, P["visual_projection.weight"])
