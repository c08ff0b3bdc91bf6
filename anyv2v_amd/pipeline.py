"""``I2VGenXLPipeline`` with the reference's call surface (seam B5; ``i2vgen-xl/pipelines/pipeline_i2vgen_xl.py:133``):
``from_pretrained``, ``.to``, ``register_modules``, ``encode_vae_video`` (:565), ``invert`` (:1197), ``__call__``
(:652), ``sample_with_pnp`` (:892) -- driving the MI355X-native UNet.

The three denoising loops (inversion :1385-1433, PnP sampling :1131-1179, CFG sampling :839-874) are the hot path.
What changed relative to the reference, all exact:
  * the inversion trajectory stays in HBM (``LatentTrajectory``); ``ddim_latents_{t}.pt`` files are still written
    (background thread, after the loop) and still readable, but no step blocks on disk or H2D;
  * no per-step ``t.item()`` sync: timesteps / DDIM coefficients live in small device tables;
  * CFG combine + scheduler step + the two permutes are one kernel reading the UNet's channels-last output;
  * a whole step (UNet forward + CFG/DDIM step) is captured once per injection state into a HIP graph and replayed.
VAE / CLIP encoders are optional components: without them the pipeline takes precomputed
``prompt_embeds`` / ``image_embeddings`` / ``image_latents`` and returns latents.
"""
from __future__ import annotations

import logging
import os
from typing import Dict, Optional, Union

import torch

from . import ops, pnp_utils
from .schedulers import DDIMScheduler
from .unet import I2VGenXLUNet, I2VGenXLUNetConfig
from .utils import LatentTrajectory, capture_hip_graph, load_ddim_latents_at_t, release_graphs

logger = logging.getLogger(__name__)


class I2VGenXLPipelineOutput:
    def __init__(self, frames):
        self.frames = frames


class StableVideoDiffusionInversionPipelineOutput:
    def __init__(self, inverted_latents):
        self.inverted_latents = inverted_latents


def _use_graphs() -> bool:
    return os.environ.get("ANYV2V_NO_GRAPH", "0") != "1"


class SourceFeatureCache:
    """Job-level exact saving for several edits of ONE clip (``configs/group_pnp_edit/group_config.json`` holds 8 edits of one
    clip): the source branch of ``sample_with_pnp`` is only a feature generator -- its v-prediction is discarded
    (``pipeline_i2vgen_xl.py:1136,1160-1162``) and it depends on the clip, its inversion and the step, not on the edit.  The first
    edit runs the three-branch steps and RECORDS, per step and injected hook site, what the other branches read from the source
    branch (Q | K of the 16 attention sites, the conv features of ``up_blocks[1].resnets[1]``); every further edit of the same
    clip REPLAYS them and runs [negative, editing] only.  Kept in HBM (~0.85 GB per fully injected step at 16 f x 512^2).

    ``signature``: what the recorded features depend on; a different one (next clip, other weights) empties the cache."""

    def __init__(self, max_bytes: Optional[int] = None):
        self.signature = None
        self.steps: Dict[tuple, Dict[str, torch.Tensor]] = {}   # (t, injection state) -> site -> features
        self.recorded_steps = self.replayed_steps = 0
        # byte budget (``ANYV2V_SOURCE_CACHE_GB``, default 96 GB of the 288): a step that would exceed it is simply not recorded and
        # runs as a three-branch step in every edit -- same results, no saving for that step
        self.max_bytes = int(float(os.environ.get("ANYV2V_SOURCE_CACHE_GB", "96")) * 2 ** 30) if max_bytes is None else int(max_bytes)
        self.skipped_steps = 0

    def store(self, t, state, feats: Dict[str, torch.Tensor]) -> bool:
        need = sum(v.numel() * v.element_size() for v in feats.values())
        if self.nbytes() + need > self.max_bytes:
            self.skipped_steps += 1
            return False
        self.steps[(int(t), tuple(state))] = {n: v.clone() for n, v in feats.items()}
        return True

    def bind(self, signature):
        if signature != self.signature:
            self.steps.clear()
            self.signature = signature

    def has(self, t, state) -> bool:
        """Features of step ``t`` recorded under exactly this injection ``state`` (which sites are on).  The same state, not a
        superset: the source branch's own arithmetic depends on it at the rounding level (on a conv-injection step its main path
        runs as a one-branch launch with its own split-K plan, at an injected attention site its Q | K | V come from a one-branch
        projection), so only a same-state record reproduces the uncached edit bit for bit."""
        return (int(t), tuple(state)) in self.steps

    def nbytes(self) -> int:
        return sum(v.numel() * v.element_size() for d in self.steps.values() for v in d.values())


class _StepEngine:
    """One denoising step = UNet forward on ``sample[B,4,F,h,w]`` + fused CFG/DDIM update of the latent slot.

    ``sample`` is a static buffer: slot ``lat_slot`` (the last one) IS the latent being denoised and is updated in
    place; ``dup_slots`` are re-filled from it after the update (CFG feeds the same latent twice); slot 0 of a PnP
    step is overwritten by the caller with the source trajectory before each step.
    """

    def __init__(self, pipe, sample, cond, b_unc, b_cond, guidance, dup_slots, shared_stem=False, batch_hint=None, lat_slots=None):
        self.pipe, self.unet = pipe, pipe.unet
        # (num, den): this engine runs a subset of another engine's branches ([negative, editing] of a three-branch edit step) and
        # must make the same launch choices -- ops.batch_hint
        self.batch_hint = batch_hint
        self.sample = sample
        self.cond = cond
        self.b_unc, self.b_cond, self.guidance = b_unc, b_cond, float(guidance)
        self.lat_slot = sample.shape[0] - 1
        # several clips inverted in one batch (``invert_clips``): every slot is a latent of its own, stepped without guidance
        self.lat_slots = None if lat_slots is None else list(lat_slots)
        self.dup_slots = list(dup_slots)
        self.coef = torch.zeros(4, dtype=torch.float32, device=sample.device)
        self.graphs: Dict[tuple, torch.cuda.CUDAGraph] = {}
        # (a frame-parallel forward contains collectives: run eagerly -- at 128 frames launch overhead is irrelevant)
        # (likewise with foreign hooks on the seams B1 / B2: torch-style code that may sync, e.g. ``t in cuda_tensor``)
        self.use_graphs = (_use_graphs() and sample.is_cuda and getattr(pipe.unet, "frame_parallel", None) is None
                           and not pnp_utils.has_foreign_hooks(pipe.unet))
        B, _, F, H, W = sample.shape
        unet = self.unet
        if not unet._packed:
            unet.pack()
        with ops.batch_hint(*(batch_hint or (1, 1))):
            self.ctx = unet._prepare_clip(B, F, H, W, cond["encoder_hidden_states"], cond["fps"], cond["image_latents"],
                                          cond["image_embeddings"])
        # CFG batches [.., negative, positive]: the last two slots hold the same latent (dup_slots) and -- checked here,
        # once -- the same image latents and fps, so the UNet may share their stem (exact; unet._forward_core)
        self._shared_stem_requested = bool(shared_stem)
        self.nosrc = None  # the [negative, editing]-only engine of a PnP edit (shares `sample[1:]`), built on demand
        self.drop_src_tail = False  # set by sample_with_pnp on its three-branch engine
        self.ctx.shared_stem = bool(shared_stem and B >= 2 and os.environ.get("ANYV2V_SHARED_STEM", "1") == "1"
                                    and torch.equal(cond["image_latents"][B - 2], cond["image_latents"][B - 1])
                                    and torch.equal(cond["fps"][B - 2], cond["fps"][B - 1]))

    def rebind(self, sample_init, cond) -> bool:
        """Point this engine (its static buffers and captured graphs) at another clip of the same geometry: the new latents go
        into ``self.sample``, the new clip's step-invariant tensors are computed by ``_prepare_clip`` and copied INTO the
        context tensors the graphs were captured on.  False if the clip cannot run on this engine (different shared-stem
        decision): the caller then builds a fresh one."""
        B, _, F, H, W = self.sample.shape
        shared = bool(self._shared_stem_requested and B >= 2 and os.environ.get("ANYV2V_SHARED_STEM", "1") == "1"
                      and torch.equal(cond["image_latents"][B - 2], cond["image_latents"][B - 1])
                      and torch.equal(cond["fps"][B - 2], cond["fps"][B - 1]))
        if shared != self.ctx.shared_stem:
            return False
        if sample_init.data_ptr() != self.sample.data_ptr():
            self.sample.copy_(sample_init)
        with ops.batch_hint(*(self.batch_hint or (1, 1))):
            fresh = self.unet._prepare_clip(B, F, H, W, cond["encoder_hidden_states"], cond["fps"], cond["image_latents"],
                                            cond["image_embeddings"])
        if fresh is not self.ctx:
            if (fresh.Sk, fresh.F) != (self.ctx.Sk, self.ctx.F):
                return False
            for name, new in vars(fresh).items():
                old = getattr(self.ctx, name, None)
                if torch.is_tensor(new) and name not in ("stats", "t_buf"):
                    if not torch.is_tensor(old) or old.shape != new.shape or old.dtype != new.dtype:
                        return False
                    old.copy_(new)
            self.ctx.key, self.ctx._keepalive = fresh.key, fresh._keepalive
            self.unet._ctx = self.ctx
        self.cond = cond
        return True

    def _body(self):
        if self.batch_hint is not None:
            self.ctx.batch_hint = self.batch_hint   # (the forward switches between the stem's and the full batch's ratio)
            with ops.batch_hint(*self.batch_hint):
                return self._body_inner()
        return self._body_inner()

    def _body_inner(self):
        # drop_src_tail: the engine's slot 0 is the PnP source branch, whose prediction the update below never reads -- the forward
        # stops computing it behind the last hook site and returns the rows of slots [1:] (unet._forward_core)
        vtok = self.unet._forward_core(self.ctx, self.sample, drop_source_tail=self.drop_src_tail)
        if self.lat_slots is not None:
            for L in self.lat_slots:
                lat = self.sample[L:L + 1]
                ops.cfg_ddim_step(vtok, -1, L, 1.0, self.coef, lat, lat)
            return
        L = self.lat_slot
        lat = self.sample[L:L + 1]
        off = 1 if vtok.shape[0] != self.sample.shape[0] * self.sample.shape[2] * self.sample.shape[3] * self.sample.shape[4] else 0
        ops.cfg_ddim_step(vtok, self.b_unc - off if self.b_unc >= 0 else self.b_unc, self.b_cond - off, self.guidance, self.coef, lat, lat)
        for s in self.dup_slots:
            self.sample[s].copy_(self.sample[L])

    def step(self, t_row: torch.Tensor, coef_row: torch.Tensor, key: tuple):
        with ops.workspace_slot(getattr(self.pipe, "ws_slot", 0)):
            self._step(t_row, coef_row, key)

    def _step(self, t_row: torch.Tensor, coef_row: torch.Tensor, key: tuple):
        self.ctx.t_buf.copy_(t_row, non_blocking=True)
        self.coef.copy_(coef_row, non_blocking=True)
        if not self.use_graphs:
            self._body()
            return
        g = self.graphs.get(key)
        if g is None:
            # warm up eagerly on the side (pure: the UNet forward does not touch the latents), then capture
            if self.batch_hint is not None:
                self.ctx.batch_hint = self.batch_hint
                with ops.batch_hint(*self.batch_hint):
                    self.unet._forward_core(self.ctx, self.sample)
            else:
                self.unet._forward_core(self.ctx, self.sample, drop_source_tail=self.drop_src_tail)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with capture_hip_graph(g):
                self._body()
            self.graphs[key] = g
            # capture does not execute: fall through to the replay below
        g.replay()


class I2VGenXLPipeline:
    def __init__(self, vae=None, text_encoder=None, tokenizer=None, image_encoder=None, feature_extractor=None,
                 unet: Optional[I2VGenXLUNet] = None, scheduler=None):
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.image_encoder, self.feature_extractor = image_encoder, feature_extractor
        self.unet, self.scheduler = unet, scheduler
        self.vae_scale_factor = 8
        self._guidance_scale = 1.0
        self._device = torch.device("cpu")
        self._engines: Dict[tuple, _StepEngine] = {}  # step engines (static buffers + HIP graphs) kept across clips, LRU
        # pacing hooks of the clip pipeline (run_group_anyv2v): ``pace_record(i, n)`` is called when edit step i of n is about to be
        # enqueued, ``pace_wait(i, n)`` before inversion step i of n -- stream events, outside the captured graphs
        self.pace_record = self.pace_wait = None
        self.ws_slot = 0   # split-K scratch buffer of this pipeline's launches (``sibling``: a second stream needs its own)
        self.source_cache: Optional[SourceFeatureCache] = None   # set (``enable_source_cache``) by multi-edit jobs

    # ------------------------------------------------------------------ construction / plumbing
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=torch.float16, variant="fp16",
                        unet_config: Optional[I2VGenXLUNetConfig] = None, random_init_seed: Optional[int] = None, **kw):
        """Loads ``<path>/unet/diffusion_pytorch_model[.fp16].safetensors`` and ``<path>/vae/...`` (diffusers key naming) when
        present.
        There is no network here: for the hub id ``"ali-vilab/i2vgen-xl"`` without a local copy, random weights of
        the exact architecture are used when ``random_init_seed`` (or ANYV2V_RANDOM_INIT_SEED) is given."""
        if torch_dtype != torch.float16:
            raise ValueError("the HIP kernels compute in fp16 (fp32 accumulate); torch_dtype must be torch.float16")
        root = str(pretrained_model_name_or_path)
        cfg_json = os.path.join(root, "unet", "config.json")
        if unet_config is None and os.path.isfile(cfg_json):
            unet_config = I2VGenXLUNetConfig.from_json(cfg_json)
        unet = I2VGenXLUNet(unet_config)
        cands = [os.path.join(root, "unet", f"diffusion_pytorch_model.{variant}.safetensors"),
                 os.path.join(root, "unet", "diffusion_pytorch_model.safetensors")]
        path = next((c for c in cands if os.path.isfile(c)), None)
        if path is not None:
            from safetensors.torch import load_file
            unet.load_state_dict(load_file(path), strict=True)
        else:
            seed = random_init_seed
            if seed is None and os.environ.get("ANYV2V_RANDOM_INIT_SEED") is not None:
                seed = int(os.environ["ANYV2V_RANDOM_INIT_SEED"])
            if seed is None:
                raise FileNotFoundError(
                    f"no UNet weights under {root!r} (expected unet/diffusion_pytorch_model.{variant}.safetensors) and no "
                    "network; pass random_init_seed= (or ANYV2V_RANDOM_INIT_SEED) to run with random weights")
            unet._random_seed = seed
        scheduler = DDIMScheduler.from_pretrained(root, subfolder="scheduler")
        pipe = cls(unet=unet, scheduler=scheduler)
        # the VAE of the checkpoint, when a local copy exists: native AutoencoderKL on the HIP kernels (same key naming)
        vcands = [os.path.join(root, "vae", f"diffusion_pytorch_model.{variant}.safetensors"),
                  os.path.join(root, "vae", "diffusion_pytorch_model.safetensors")]
        vpath = next((c for c in vcands if os.path.isfile(c)), None)
        if vpath is not None:
            from safetensors.torch import load_file

            from .encoders import NativeVAE
            pipe.vae = NativeVAE(state_dict=load_file(vpath))
        if os.path.isdir(os.path.join(root, "text_encoder")):
            # CLIP towers of a local checkpoint on the HIP kernels (anyv2v_amd.clip)
            from .encoders import attach_native_clip_encoders
            attach_native_clip_encoders(pipe, root)
        return pipe

    def sibling(self, ws_slot: int = 1):
        """A second pipeline object around the SAME components (no weights copied) with its own scheduler slot, step engines,
        conditioning cache and scratch buffer: ``run_group_anyv2v`` edits clip k on one stream with it while this pipeline
        inverts clip k + 1 on another."""
        p = type(self)(vae=self.vae, text_encoder=self.text_encoder, tokenizer=self.tokenizer, image_encoder=self.image_encoder,
                       feature_extractor=self.feature_extractor, unet=self.unet, scheduler=self.scheduler)
        p._device, p.ws_slot = self._device, int(ws_slot)
        return p

    def to(self, device):
        device = torch.device(device)
        self._device = device
        self.unet.to(device)
        seed = getattr(self.unet, "_random_seed", None)
        if seed is not None:
            init_random_weights_(self.unet, seed)
            self.unet._random_seed = None
        for m in (self.vae, self.text_encoder, self.image_encoder):
            if m is not None and hasattr(m, "to"):
                m.to(device)
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

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    def progress_bar(self, iterable=None, total=None):
        return iterable

    def maybe_free_model_hooks(self):
        pass

    # ------------------------------------------------------------------ input checks (pipeline_i2vgen_xl.py:483-530)
    def check_inputs(self, prompt, image, height, width, negative_prompt=None, prompt_embeds=None,
                     negative_prompt_embeds=None):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `prompt`: {prompt} and `prompt_embeds`: {prompt_embeds}. Please make sure to"
                             " only forward one of the two.")
        elif prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        elif prompt is not None and (not isinstance(prompt, str) and not isinstance(prompt, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `negative_prompt`: {negative_prompt} and `negative_prompt_embeds`:"
                             f" {negative_prompt_embeds}. Please make sure to only forward one of the two.")
        if prompt_embeds is not None and negative_prompt_embeds is not None:
            if prompt_embeds.shape != negative_prompt_embeds.shape:
                raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, but"
                                 f" got: `prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds`"
                                 f" {negative_prompt_embeds.shape}.")

    # ------------------------------------------------------------------ encoders (optional components, F1)
    def _need(self, name):
        m = getattr(self, name)
        if m is None:
            raise RuntimeError(
                f"pipeline component `{name}` is not loaded (VAE/CLIP are outside the HIP hot path and no pretrained weights "
                "exist offline); pass the precomputed tensors instead (prompt_embeds / negative_prompt_embeds / "
                "image_embeddings / image_latents / latents, output_type='latent')")
        return m

    def encode_prompt(self, prompt, device, num_videos_per_prompt=1, negative_prompt=None, prompt_embeds=None,
                      negative_prompt_embeds=None, lora_scale=None, clip_skip=None):
        if prompt_embeds is None:
            enc, tok = self._need("text_encoder"), self._need("tokenizer")
            prompt_embeds = _clip_text(enc, tok, prompt, device, clip_skip)
        if self.do_classifier_free_guidance and negative_prompt_embeds is None:
            enc, tok = self._need("text_encoder"), self._need("tokenizer")
            n = negative_prompt if negative_prompt is not None else ""
            if isinstance(n, str):
                n = [n] * prompt_embeds.shape[0]
            negative_prompt_embeds = _clip_text(enc, tok, n, device, clip_skip)
        return prompt_embeds, negative_prompt_embeds

    def prepare_image_latents_from_first_frame_latent(self, first_latent, num_frames):
        """``prepare_image_latents`` (:532-562) after the VAE: first-frame latent + (F-1) frame-position planes."""
        il = first_latent.unsqueeze(2)  # [b,4,1,h,w]
        planes = [torch.ones_like(il[:, :, :1]) * ((i + 1) / (num_frames - 1)) for i in range(num_frames - 1)]
        if planes:
            il = torch.cat([il] + planes, dim=2)
        return il

    def prepare_image_latents(self, image, device, num_frames, num_videos_per_prompt=1):
        """``pipeline_i2vgen_xl.py:532-562``: a pre-processed first frame [1, 3, H, W] in [-1, 1] -> sampled, scaled VAE latent, the
        frame-position planes behind it, doubled under classifier-free guidance."""
        if num_videos_per_prompt not in (None, 1):
            raise ValueError("one clip per call (num_videos_per_prompt = 1)")
        first = self._need("vae").encode_pixels(image, device)
        il = self.prepare_image_latents_from_first_frame_latent(first, num_frames)
        return torch.cat([il] * 2) if self.do_classifier_free_guidance else il

    def prepare_extra_step_kwargs(self, generator, eta):
        """``:466-482``: ``eta`` / ``generator`` for a scheduler whose ``step`` takes them (the forward DDIM scheduler takes both, the
        inverse one neither)."""
        import inspect
        params = set(inspect.signature(self.scheduler.step).parameters)
        extra = {}
        if "eta" in params:
            extra["eta"] = eta
        if "generator" in params:
            extra["generator"] = generator
        return extra

    # memory savers of the reference pipeline (``:192-222``): decoding already runs ``decode_chunk_size`` frames at a time, and a 512^2
    # frame is far below anything that needs tiles on this device -- both switches are accepted and change nothing (the reference's
    # tiled decode blends tile seams, i.e. differs from its own untiled output; this build always gives the untiled one)
    def enable_vae_slicing(self):
        self._vae_slicing = True

    def disable_vae_slicing(self):
        self._vae_slicing = False

    def enable_vae_tiling(self):
        logger.warning("enable_vae_tiling: frames are decoded whole on this device; the output equals the reference's UNTILED decode")
        self._vae_tiling = True

    def disable_vae_tiling(self):
        self._vae_tiling = False

    def enable_freeu(self, s1, s2, b1, b2):
        """``:623-648`` (FreeU re-weights the decoder's skip / backbone features): not built -- AnyV2V never enables it."""
        raise NotImplementedError("FreeU is not supported by the native UNet")

    def disable_freeu(self):
        pass

    def encode_vae_video(self, video, device, height=576, width=1024):
        vae = self._need("vae")
        return vae.encode_video(video, device, height, width)

    def decode_latents(self, latents, decode_chunk_size=None):
        vae = self._need("vae")
        fp = getattr(self.unet, "frame_parallel", None)
        if fp is not None and latents.shape[2] % fp.world == 0:
            # frame-parallel clip: frames decode independently (``decode_chunk_size=1`` in the reference) -- every rank
            # decodes its own frames, one all_gather of the decoded frames (12 MiB per 16 frames at 512^2)
            f0, f1 = fp.frames(latents.shape[2])
            mine = vae.decode_video(latents[:, :, f0:f1].contiguous(), decode_chunk_size)
            return fp.gather_video(mine)
        return vae.decode_video(latents, decode_chunk_size)

    def prepare_latents(self, batch_size, num_channels_latents, num_frames, height, width, dtype, device, generator,
                        latents=None):
        shape = (batch_size, num_channels_latents, num_frames, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if latents is None:
            latents = torch.randn(shape, generator=generator, dtype=torch.float32).to(device=device, dtype=dtype)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    @staticmethod
    def _single_clip(prompt_embeds, num_videos_per_prompt):
        """The loops drive ONE clip per call (``nb`` batch slots = its CFG / PnP branches).  The reference computes a real
        batch for a list prompt; here that would silently combine the negative branch of prompt 0 with the positive branch
        of prompt 1, so it is refused."""
        if num_videos_per_prompt not in (None, 1):
            raise ValueError(f"num_videos_per_prompt={num_videos_per_prompt} is not supported: one clip per call (shard clips "
                             "over processes / GPUs instead, anyv2v_amd.parallel)")
        if prompt_embeds.shape[0] != 1:
            raise ValueError(f"got a batch of {prompt_embeds.shape[0]} prompts: one clip per call (the batch dimension holds the "
                             "CFG / PnP branches of that clip)")

    def _forward_sampler_options(self, eta, cross_attention_kwargs):
        """``__call__`` / ``sample_with_pnp`` (``pipeline_i2vgen_xl.py:798,868`` / ``:1095,1173``): ``eta`` goes to ``scheduler.step`` when
        that takes it -- the forward DDIM scheduler does (stochastic DDIM), the inverse one does not (``invert`` drops it, as here).
        The fused guidance + DDIM step of the step engines is the deterministic one; AnyV2V never sets either option, and silently
        ignoring them would change the sample."""
        if eta not in (None, 0, 0.0) and isinstance(self.scheduler, DDIMScheduler):
            raise ValueError(f"eta={eta}: the step engines run deterministic DDIM (eta = 0) only")
        if cross_attention_kwargs:
            raise ValueError("cross_attention_kwargs (LoRA scale, ...) are not supported by the native attention processors")

    # ------------------------------------------------------------------ conditioning assembly
    def _conditioning(self, prompt, image, height, width, num_frames, negative_prompt, prompt_embeds,
                      negative_prompt_embeds, target_fps, clip_skip, image_embeddings, image_latents):
        """Returns per-branch lists: cond branch tensors and (when CFG) uncond ones, following :1014-1101."""
        device = self._execution_device
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, device, 1, negative_prompt, prompt_embeds=prompt_embeds,
            negative_prompt_embeds=negative_prompt_embeds, clip_skip=clip_skip)
        if image_embeddings is None:
            image_embeddings = _clip_image(self._need("image_encoder"), self._need("feature_extractor"), image, width, device)
        if image_latents is None:
            first = self._need("vae").encode_image(image, device, height, width)
            image_latents = self.prepare_image_latents_from_first_frame_latent(first, num_frames)
        return (prompt_embeds.to(device, torch.float16), None if negative_prompt_embeds is None else
                negative_prompt_embeds.to(device, torch.float16), image_embeddings.to(device, torch.float16),
                image_latents.to(device, torch.float16))

    def enable_source_cache(self, on: bool = True):
        """Several edits of one clip (a group job, the front-end called repeatedly): keep the source branch's injected features of
        the clip in HBM across ``sample_with_pnp`` calls -- ``SourceFeatureCache``.  Off by default: a single edit gains nothing."""
        self.source_cache = SourceFeatureCache() if on else None
        return self.source_cache

    def _engine(self, tag, sample, cond, **kw) -> _StepEngine:
        """A step engine for this loop: a new one, or -- for the next clip of the same geometry in a multi-clip job -- the one
        captured for the previous clip, re-pointed at the new clip (``_StepEngine.rebind``): graph capture and its warm-up
        forward (~0.25 s per 16 x 512^2 clip) are paid once per process instead of once per clip.  ANYV2V_ENGINE_CACHE=0
        switches it off."""
        if os.environ.get("ANYV2V_ENGINE_CACHE", "1") != "1" or getattr(self.unet, "frame_parallel", None) is not None:
            return _StepEngine(self, sample, cond, **kw)
        if not self.unet._packed:
            self.unet.pack()  # (weights loaded / moved since the last call: new packed tensors, engines of the old ones are stale)
        key = (tag, self.unet._pack_gen, tuple(sample.shape), str(sample.device), tuple(cond["encoder_hidden_states"].shape), kw.get("b_unc"),
               kw.get("b_cond"), float(kw.get("guidance")), tuple(kw.get("dup_slots")), bool(kw.get("shared_stem", False)),
               _use_graphs(), pnp_utils.has_foreign_hooks(self.unet), id(self.unet), kw.get("lat_slots"))
        eng = self._engines.pop(key, None)
        if eng is None or not eng.rebind(sample, cond):
            eng = _StepEngine(self, sample, cond, **kw)
        # Every engine pins a private graph pool holding a whole forward's activations (and a nested one for the source-free
        # steps), so the cache is small and evicts eagerly: one engine per loop kind (a new guidance value or geometry replaces
        # the old engine of that loop -- a slider in a long-lived front-end must not accumulate pools), at most
        # ANYV2V_ENGINE_CACHE_MAX (3 = inversion + CFG + PnP) overall; an evicted engine drops its graphs and returns the memory.
        evict = [k for k in self._engines if k[0] == tag]
        self._engines[key] = eng  # most recently used last
        limit = max(1, int(os.environ.get("ANYV2V_ENGINE_CACHE_MAX", "3")))
        evict += [k for k in list(self._engines)[: max(0, len(self._engines) - len(evict) - limit)] if k not in evict]
        for k in evict:
            old = self._engines.pop(k)
            release_graphs(old.graphs)   # (parked, not destroyed, while another engine / thread is capturing: utils.release_graphs)
            if old.nosrc is not None:
                release_graphs(old.nosrc.graphs)
                old.nosrc = None
        if evict and sample.is_cuda:
            torch.cuda.empty_cache()
        return eng

    # ------------------------------------------------------------------ A1: DDIM inversion (:1197-1451)
    @torch.no_grad()
    def invert(self, prompt=None, image=None, height: Optional[int] = 704, width: Optional[int] = 1280,
               target_fps: Optional[int] = 16, num_frames: int = 16, num_inference_steps: int = 50,
               guidance_scale: float = 9.0, negative_prompt=None, eta: float = 0.0, num_videos_per_prompt: Optional[int] = 1,
               decode_chunk_size: Optional[int] = 1, generator=None, latents: Optional[torch.Tensor] = None,
               prompt_embeds=None, negative_prompt_embeds=None, output_type: Optional[str] = "pil",
               return_dict: bool = True, cross_attention_kwargs=None, clip_skip: Optional[int] = 1,
               output_dir: Optional[str] = None, image_embeddings=None, image_latents=None,
               return_trajectory: bool = False, background_save: bool = False):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, image, height, width, negative_prompt, prompt_embeds, negative_prompt_embeds)
        device = self._execution_device
        self._guidance_scale = guidance_scale
        pnp_utils.clear_time(self)  # inversion never injects (stage 1 of the reference runs without hooks)
        pe, npe, ie, il = self._conditioning(prompt, image, height, width, num_frames, negative_prompt, prompt_embeds,
                                             negative_prompt_embeds, target_fps, clip_skip, image_embeddings, image_latents)
        self._single_clip(pe, num_videos_per_prompt)
        cfg_on = self.do_classifier_free_guidance
        if cfg_on:  # [uncond, cond] (:1329-1337); zero negative image embedding (:437-439)
            ehs = torch.cat([npe, pe])
            ie_all = torch.cat([torch.zeros_like(ie), ie])
            il_all = torch.cat([il, il])
        else:
            ehs, ie_all, il_all = pe, ie, il
        nb = ehs.shape[0]
        fps = torch.tensor([target_fps] * nb, device=device)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        latents = self.prepare_latents(1, self.unet.config.in_channels, num_frames, height, width, torch.float16, device,
                                       generator, latents).to(torch.float16)
        sample = latents.repeat(nb, 1, 1, 1, 1).contiguous()
        cond = dict(encoder_hidden_states=ehs.contiguous(), fps=fps, image_latents=il_all.contiguous(),
                    image_embeddings=ie_all.contiguous())
        eng = self._engine("inv", sample, cond, b_unc=0 if cfg_on else -1, b_cond=nb - 1, guidance=guidance_scale,
                           dup_slots=range(nb - 1))
        sample = eng.sample  # (a cached engine keeps its own static buffer; the latents were copied into it)
        ts = [int(t) for t in timesteps.tolist()]
        t_table = torch.tensor(ts, dtype=torch.float32, device=device)[:, None].expand(-1, nb).contiguous()
        coef_table = self.scheduler.coefficient_table(ts, device)
        traj = LatentTrajectory()
        for i, t in enumerate(ts):
            if self.pace_wait is not None:
                self.pace_wait(i, len(ts))
            eng.step(t_table[i], coef_table[i], key=("inv",))
            traj[t] = sample[nb - 1:nb].clone()
        if output_dir is not None:
            # ddim_latents_{t}.pt, reference format.  Complete when invert() returns (as in the reference, which writes
            # inside the loop) unless the caller opts into the background writer (the CLI runner does; every reader in
            # anyv2v_amd.utils joins it first)
            traj.save(output_dir, background=background_save)
            logger.info(f"saving noisy latents for {len(ts)} timesteps to {output_dir}")
        self._last_trajectory = traj
        inverted = torch.stack([traj[t] for t in reversed(ts)], 1)  # [1, n, 4, F, h, w] (:1436)
        if return_trajectory:
            return traj
        if not return_dict:
            return inverted
        return StableVideoDiffusionInversionPipelineOutput(inverted_latents=inverted)

    @torch.no_grad()
    def invert_clips(self, clips, height: int, width: int, num_frames: int = 16, num_inference_steps: int = 50, target_fps: int = 16,
                     clip_skip: Optional[int] = 1, output_dirs=None, background_save: bool = False):
        """Several clips inverted in ONE batch (guidance 1, the inversion configuration of the reference's runner): ``clips`` = list of
        dicts ``prompt``, ``image`` (first frame), ``latents`` [1, 4, F, h, w] (``encode_vae_video``), optionally ``negative_prompt``.
        The UNet forward runs over B = len(clips) rows -- each row its own clip with its own conditioning --, so every weight is read
        once per step for all of them and the low-resolution launches carry B times the rows: an inversion-bound job (the template's
        500 steps) gains what a B = 1 launch loses against a B = 3 one.  Returns one ``LatentTrajectory`` per clip; the numbers
        differ from ``invert`` at rounding level (other launch plans at other row counts), not bit for bit."""
        device = self._execution_device
        self._guidance_scale = 1.0
        pnp_utils.clear_time(self)
        n = len(clips)
        pes, ies, ils, lats = [], [], [], []
        for c in clips:
            pe, _npe, ie, il = self._conditioning(c.get("prompt", ""), c["image"], height, width, num_frames, c.get("negative_prompt"), None, None,
                                                  target_fps, clip_skip, None, None)
            self._single_clip(pe, 1)
            pes.append(pe)
            ies.append(ie)
            ils.append(il)
            lats.append(self.prepare_latents(1, self.unet.config.in_channels, num_frames, height, width, torch.float16, device, None,
                                             c["latents"]).to(torch.float16))
        sample = torch.cat(lats).contiguous()
        cond = dict(encoder_hidden_states=torch.cat(pes).contiguous(), fps=torch.tensor([target_fps] * n, device=device),
                    image_latents=torch.cat(ils).contiguous(), image_embeddings=torch.cat(ies).contiguous())
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        ts = [int(t) for t in self.scheduler.timesteps.tolist()]
        eng = self._engine(f"inv{n}", sample, cond, b_unc=-1, b_cond=n - 1, guidance=1.0, dup_slots=(), lat_slots=tuple(range(n)))
        sample = eng.sample
        t_table = torch.tensor(ts, dtype=torch.float32, device=device)[:, None].expand(-1, n).contiguous()
        coef_table = self.scheduler.coefficient_table(ts, device)
        trajs = [LatentTrajectory() for _ in range(n)]
        for i, t in enumerate(ts):
            eng.step(t_table[i], coef_table[i], key=("inv",))
            for k in range(n):
                trajs[k][t] = sample[k:k + 1].clone()
        if output_dirs is not None:
            for tr, d in zip(trajs, output_dirs):
                if d is not None:
                    tr.save(d, background=background_save)
        return trajs

    # ------------------------------------------------------------------ A3: plain CFG sampling (:652-888)
    @torch.no_grad()
    def __call__(self, prompt=None, image=None, height: Optional[int] = 704, width: Optional[int] = 1280,
                 target_fps: Optional[int] = 16, num_frames: int = 16, num_inference_steps: int = 50,
                 guidance_scale: float = 9.0, negative_prompt=None, eta: float = 0.0, num_videos_per_prompt: Optional[int] = 1,
                 decode_chunk_size: Optional[int] = 1, generator=None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds=None, negative_prompt_embeds=None, output_type: Optional[str] = "pil",
                 return_dict: bool = True, cross_attention_kwargs=None, clip_skip: Optional[int] = 1,
                 ddim_init_latents_t_idx: Optional[int] = 1, image_embeddings=None, image_latents=None,
                 latents_trace: Optional[dict] = None):
        """``latents_trace``: optional dict, filled with t -> latents after the step at t (drift reports; not a reference
        argument)."""
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, image, height, width, negative_prompt, prompt_embeds, negative_prompt_embeds)
        device = self._execution_device
        self._guidance_scale = guidance_scale
        pe, npe, ie, il = self._conditioning(prompt, image, height, width, num_frames, negative_prompt, prompt_embeds,
                                             negative_prompt_embeds, target_fps, clip_skip, image_embeddings, image_latents)
        self._single_clip(pe, num_videos_per_prompt)
        self._forward_sampler_options(eta, cross_attention_kwargs)
        cfg_on = self.do_classifier_free_guidance
        pnp_utils.clear_time(self)  # plain CFG sampling (DDIM reconstruction) runs hook-free
        if cfg_on:
            ehs, ie_all, il_all = torch.cat([npe, pe]), torch.cat([torch.zeros_like(ie), ie]), torch.cat([il, il])
        else:
            ehs, ie_all, il_all = pe, ie, il
        nb = ehs.shape[0]
        fps = torch.tensor([target_fps] * nb, device=device)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        self.scheduler.timesteps = self.scheduler.timesteps[ddim_init_latents_t_idx:]  # :813
        ts = [int(t) for t in self.scheduler.timesteps.tolist()]
        latents = self.prepare_latents(1, self.unet.config.in_channels, num_frames, height, width, torch.float16, device,
                                       generator, latents).to(torch.float16)
        sample = latents.repeat(nb, 1, 1, 1, 1).contiguous()
        cond = dict(encoder_hidden_states=ehs.contiguous(), fps=fps, image_latents=il_all.contiguous(),
                    image_embeddings=ie_all.contiguous())
        eng = self._engine("cfg", sample, cond, b_unc=0 if cfg_on else -1, b_cond=nb - 1, guidance=guidance_scale,
                           dup_slots=range(nb - 1), shared_stem=cfg_on)
        sample = eng.sample
        t_table = torch.tensor(ts, dtype=torch.float32, device=device)[:, None].expand(-1, nb).contiguous()
        coef_table = self.scheduler.coefficient_table(ts, device)
        for i, t in enumerate(ts):
            eng.step(t_table[i], coef_table[i], key=("cfg",))
            if latents_trace is not None:
                latents_trace[t] = sample[nb - 1:nb].clone()
        return self._finish(sample[nb - 1:nb].clone(), output_type, decode_chunk_size, return_dict)

    # ------------------------------------------------------------------ A2: PnP edit (:892-1193)
    @torch.no_grad()
    def sample_with_pnp(self, prompt=None, image=None, height: Optional[int] = 704, width: Optional[int] = 1280,
                        target_fps: Optional[int] = 16, num_frames: int = 16, num_inference_steps: int = 50,
                        guidance_scale: float = 9.0, negative_prompt=None, eta: float = 0.0,
                        num_videos_per_prompt: Optional[int] = 1, decode_chunk_size: Optional[int] = 1, generator=None,
                        latents: Optional[torch.Tensor] = None, prompt_embeds=None, negative_prompt_embeds=None,
                        output_type: Optional[str] = "pil", return_dict: bool = True, cross_attention_kwargs=None,
                        clip_skip: Optional[int] = 1, ddim_init_latents_t_idx: Optional[int] = 1,
                        ddim_inv_latents_path: Union[str, LatentTrajectory, None] = None, ddim_inv_prompt=None,
                        ddim_inv_1st_frame=None, image_embeddings=None, image_latents=None,
                        ddim_inv_prompt_embeds=None, ddim_inv_image_embeddings=None, ddim_inv_image_latents=None,
                        latents_trace: Optional[dict] = None):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, image, height, width, negative_prompt, prompt_embeds, negative_prompt_embeds)
        if isinstance(prompt, list):
            assert len(ddim_inv_prompt) == len(prompt)  # :1000
        device = self._execution_device
        self._guidance_scale = guidance_scale
        pe, npe, ie, il = self._conditioning(prompt, image, height, width, num_frames, negative_prompt, prompt_embeds,
                                             negative_prompt_embeds, target_fps, clip_skip, image_embeddings, image_latents)
        self._single_clip(pe, num_videos_per_prompt)
        self._forward_sampler_options(eta, cross_attention_kwargs)
        # source (ddim inversion) branch: its own prompt, first frame, positive image embedding (:1027-1091)
        if ddim_inv_prompt_embeds is None:
            ddim_inv_prompt_embeds = _clip_text(self._need("text_encoder"), self._need("tokenizer"), ddim_inv_prompt,
                                                device, clip_skip)
        if ddim_inv_image_embeddings is None:
            ddim_inv_image_embeddings = _clip_image(self._need("image_encoder"), self._need("feature_extractor"),
                                                    ddim_inv_1st_frame, width, device)
        if ddim_inv_image_latents is None:
            first = self._need("vae").encode_image(ddim_inv_1st_frame, device, height, width)
            ddim_inv_image_latents = self.prepare_image_latents_from_first_frame_latent(first, num_frames)
        spe = ddim_inv_prompt_embeds.to(device, torch.float16)
        sie = ddim_inv_image_embeddings.to(device, torch.float16)
        sil = ddim_inv_image_latents.to(device, torch.float16)
        cfg_on = self.do_classifier_free_guidance
        if cfg_on:  # order: [ddim_inversion, negative, editing] (:1044,1093-1094)
            ehs = torch.cat([spe, npe, pe])
            ie_all = torch.cat([sie, torch.zeros_like(ie), ie])
            il_all = torch.cat([sil, il, il])
        else:
            ehs, ie_all, il_all = torch.cat([spe, pe]), torch.cat([sie, ie]), torch.cat([sil, il])
        nb = ehs.shape[0]
        fps = torch.tensor([target_fps] * nb, device=device)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        self.scheduler.timesteps = self.scheduler.timesteps[ddim_init_latents_t_idx:]  # :1105
        ts = [int(t) for t in self.scheduler.timesteps.tolist()]
        latents = self.prepare_latents(1, self.unet.config.in_channels, num_frames, height, width, torch.float16, device,
                                       generator, latents).to(torch.float16)
        sample = latents.repeat(nb, 1, 1, 1, 1).contiguous()
        cond = dict(encoder_hidden_states=ehs.contiguous(), fps=fps, image_latents=il_all.contiguous(),
                    image_embeddings=ie_all.contiguous())
        # the source branch's prediction is never read (:1136,1160-1162): its forward may stop behind the last hook site (exact)
        drop_tail = cfg_on and nb == 3 and os.environ.get("ANYV2V_DROP_SRC_TAIL", "1") == "1"
        eng = self._engine("pnp-droptail" if drop_tail else "pnp", sample, cond, b_unc=1 if cfg_on else -1, b_cond=nb - 1,
                           guidance=guidance_scale, dup_slots=range(1, nb - 1), shared_stem=cfg_on)
        eng.drop_src_tail = drop_tail
        sample = eng.sample
        # source trajectory resident in HBM (in-memory hand-off from invert(), or read once from the reference's files)
        if isinstance(ddim_inv_latents_path, LatentTrajectory):
            traj = ddim_inv_latents_path
        else:
            traj = LatentTrajectory.load(ddim_inv_latents_path, device=device, timesteps=ts)
        t_table = torch.tensor(ts, dtype=torch.float32, device=device)[:, None].expand(-1, nb).contiguous()
        coef_table = self.scheduler.coefficient_table(ts, device)
        # Exact work elimination (SURVEY A.5 probe "branches 1,2 of a B=3 forward equal a B=2 forward"): on a step that lies
        # outside every injection schedule nothing reads the source branch, so the step runs on the [negative, editing]
        # slots only.  Both engines share the `sample` storage, so they can alternate freely.
        skip_src = cfg_on and nb == 3 and os.environ.get("ANYV2V_SRC_SKIP", "1") == "1"
        eng_nosrc = None
        nosrc_bound = False
        # multi-edit job: record / replay what the injected sites read from the source branch (SourceFeatureCache)
        cache = self.source_cache if (skip_src and not pnp_utils.has_foreign_hooks(self.unet)
                                      and getattr(self.unet, "frame_parallel", None) is None) else None
        sites = pnp_utils.injection_sites(self) if cache is not None else []
        if cache is not None:
            fp_ = lambda x: (tuple(x.shape), float(x.float().sum()), float(x.float().abs().sum()))
            src = ("object", traj.serial) if isinstance(ddim_inv_latents_path, LatentTrajectory) else ("files", os.path.abspath(str(ddim_inv_latents_path)))
            cache.bind((src, tuple(ts), fp_(load_ddim_latents_at_t(ts[0], traj)), fp_(spe), fp_(sie), fp_(sil), int(target_fps),
                        self.unet._pack_gen, id(self.unet), tuple(latents.shape)))
            if not hasattr(eng, "site_bufs"):
                # rows of ONE branch at the site's level: up_blocks[1] works at 1/16 of the 64x64 level's pixels, [2] at 1/4, [3] at 1/1
                full = num_frames * (height // self.vae_scale_factor) * (width // self.vae_scale_factor)
                level_div = {"1": 16, "2": 4, "3": 1}
                eng.site_bufs = {name: torch.empty((full // level_div[name.split(".up")[1][0]], cols), dtype=torch.float16, device=device)
                                 for name, _obj, cols in sites}

        def set_io(mode, names):
            for name, obj, _ in sites:
                obj.src_io = (mode, eng.site_bufs[name]) if (mode is not None and name in names) else None

        for i, t in enumerate(ts):
            if self.pace_record is not None:
                self.pace_record(i, len(ts))
            pnp_utils.register_time(self, t)  # host-side only: python int, no device sync (:1143)
            state = pnp_utils.injection_state(self)
            if any(state) and nb != 3:
                # the hooks slice the batch in thirds (pnp_utils.py:111,166,272); with guidance_scale <= 1 the reference
                # builds a 2-way batch and silently injects the wrong rows -- refuse instead
                raise ValueError("PnP feature injection needs classifier-free guidance (guidance_scale > 1): the hooks "
                                 "assume the 3-way batch [source, negative, editing]")
            if skip_src and not any(state):
                if not nosrc_bound:  # built once per engine (it lives on `sample[1:]`), re-pointed at this clip once per call
                    cond2 = {k: v[1:].contiguous() for k, v in cond.items()}
                    if eng.nosrc is None or not eng.nosrc.rebind(sample[1:], cond2):
                        eng.nosrc = _StepEngine(self, sample[1:], cond2, b_unc=0, b_cond=1, guidance=guidance_scale,
                                                dup_slots=[0], shared_stem=True, batch_hint=(3, 2))
                    eng_nosrc, nosrc_bound = eng.nosrc, True
                eng_nosrc.step(t_table[i, 1:], coef_table[i], key=("pnp-nosrc",))
            elif cache is not None and cache.has(t, state):
                # replay: the source features of this step are in HBM -- [negative, editing] only
                names = [n for (n, _, _), on in zip(sites, state) if on]
                if not nosrc_bound:
                    cond2 = {k: v[1:].contiguous() for k, v in cond.items()}
                    if eng.nosrc is None or not eng.nosrc.rebind(sample[1:], cond2):
                        eng.nosrc = _StepEngine(self, sample[1:], cond2, b_unc=0, b_cond=1, guidance=guidance_scale,
                                                dup_slots=[0], shared_stem=True, batch_hint=(3, 2))
                    eng_nosrc, nosrc_bound = eng.nosrc, True
                for n in names:
                    eng.site_bufs[n].copy_(cache.steps[(int(t), state)][n], non_blocking=True)
                set_io("replay", names)
                eng_nosrc.step(t_table[i, 1:], coef_table[i], key=("pnp-replay",) + state)
                set_io(None, ())
                cache.replayed_steps += 1
            else:
                sample[0].copy_(load_ddim_latents_at_t(t, traj).to(device=device, dtype=torch.float16)[0], non_blocking=True)
                if cache is not None:
                    names = [n for (n, _, _), on in zip(sites, state) if on]
                    set_io("record", names)
                    eng.step(t_table[i], coef_table[i], key=("pnp-record",) + state)
                    set_io(None, ())
                    cache.recorded_steps += bool(cache.store(t, state, {n: eng.site_bufs[n] for n in names}))
                else:
                    eng.step(t_table[i], coef_table[i], key=("pnp",) + state)
            if latents_trace is not None:
                latents_trace[t] = sample[nb - 1:nb].clone()
        return self._finish(sample[nb - 1:nb].clone(), output_type, decode_chunk_size, return_dict)

    def _finish(self, latents, output_type, decode_chunk_size, return_dict):
        if output_type == "latent":
            return I2VGenXLPipelineOutput(frames=latents)
        if output_type not in ("pil", "np", "pt"):
            raise ValueError(f"{output_type} does not exist. Please choose one of ['np', 'pt', 'pil]")     # (``tensor2vid``, :94-95)
        video = self.decode_latents(latents, decode_chunk_size=decode_chunk_size)          # [1, 3, F, H, W] in [-1, 1]
        frames = tensor2vid(video, self._need("vae"), output_type)      # (the runners index ``.frames[0]``, ``run_group_ddim_inversion.py:77``)
        if not return_dict:
            return (frames,)
        return I2VGenXLPipelineOutput(frames=frames)


def tensor2vid(video: torch.Tensor, processor=None, output_type: str = "np"):
    """``pipeline_i2vgen_xl.py:79-97`` over ``VaeImageProcessor.postprocess``: [b, 3, f, H, W] in [-1, 1] -> "pil": one list of PIL frames
    per video ((x / 2 + 0.5) * 255 rounded); "pt": [b, f, 3, H, W] in [0, 1]; "np": the same as float32 numpy [b, f, H, W, 3].
    ``processor``: anything with ``to_pil(video[1, 3, f, H, W])`` (the VAE adapters quantise where the tensor lives) or None."""
    if output_type == "pil":
        def one(v):
            if processor is not None and hasattr(processor, "to_pil"):
                return processor.to_pil(v)
            x = ((v[0].permute(1, 2, 3, 0).float() + 1.0) * 127.5).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
            from PIL import Image
            return [Image.fromarray(fr) for fr in x]
        return [one(video[b:b + 1]) for b in range(video.shape[0])]
    if output_type not in ("np", "pt"):
        raise ValueError(f"{output_type} does not exist. Please choose one of ['np', 'pt', 'pil]")
    frames = (video.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 1, 3, 4)
    return frames.permute(0, 1, 3, 4, 2).cpu().numpy() if output_type == "np" else frames


# ---------------------------------------------------------------------------------------------------------------
def init_random_weights_(unet: I2VGenXLUNet, seed: int):
    """Random weights of the exact architecture, generated on the device (no pretrained weights offline).
    N(0, 1/fan_in) matrices, N(0, 0.02) biases, N(1, 0.05) norm gains -- same law as oracle.random_state_dict."""
    g = torch.Generator(device=unet.device).manual_seed(seed)
    for name, p in unet.named_parameters():
        shp = p.shape
        if name.endswith(".bias"):
            t = torch.randn(shp, generator=g, device=p.device) * 0.02
        elif p.dim() == 1:
            t = 1.0 + torch.randn(shp, generator=g, device=p.device) * 0.05
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = torch.randn(shp, generator=g, device=p.device) / (fan_in ** 0.5)
        p.data.copy_(t.to(torch.float16))
    unet._packed = False


def _clip_text(text_encoder, tokenizer, prompt, device, clip_skip):
    """``encode_prompt`` (:224-409) through the component interface of ``anyv2v_amd.encoders``."""
    if isinstance(prompt, str):
        prompt = [prompt]
    return text_encoder.encode(prompt, device, clip_skip)


def _clip_image(image_encoder, feature_extractor, image, width, device):
    """``_encode_image`` (:411-441) on the centre-cropped, 224x224 bilinear-resized frame (:1051-1055)."""
    return image_encoder.encode(image, width, device)
