/*
 * anyv2v_hip.h -- C ABI of libanyv2v_hip.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * AnyV2V I2VGen-XL DDIM-inversion + PnP-edit hot path.
 *
 * The reference (TIGER-AI-Lab/AnyV2V) is pure Python and has no FFI layer; its hot path sits behind
 * Python protocol seams.  Each entry point below names the reference call it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp16 ("half") data unless the name says f32 / i32 / i64;
 *   - activations are channels-last token matrices  X[(b f) (h w), C]  (row-major, leading dim in elements);
 *   - `stream` is a hipStream_t (0 = default stream); kernels are enqueued, never synchronised;
 *   - nothing allocates: the caller owns all memory (PyTorch caching allocator in the Python host);
 *   - return value: 0 = ok, <0 = ANYV2V_E* (message via anyv2v_last_error()), >0 = hipError_t.
 */
#ifndef ANYV2V_HIP_H
#define ANYV2V_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANYV2V_OK 0
#define ANYV2V_EINVAL (-1)   /* bad shape / alignment / null pointer            */
#define ANYV2V_EUNSUPPORTED (-2) /* shape outside what the fast kernels cover   */

/* ---- gather-GEMM ------------------------------------------------------------------------------
 * C[m, n] = epilogue( sum_{tap} sum_{k<K} A[row(m,tap), k] * W[n, tap*K + k] ),  fp16 in, fp32 acc.
 * A is the channel-concatenation [A0 | A1] (K = C0 + C1) -- the skip `torch.cat` of the up blocks
 * folded into the K loop.
 *   mode 0  LINEAR    : 1 tap, row = m.                     torch Linear / 1x1 conv
 *                       (attn.to_q/to_k/to_v/to_out[0]  i2vgen-xl/pnp_utils.py:175,182-183,216;
 *                        conv_shortcut pnp_utils.py:117-122; time_emb_proj pnp_utils.py:81-88)
 *   mode 1  CONV2D3x3 : 9 taps, pad 1, stride 1|2, optional nearest x2 upsample folded into the gather
 *                       (conv1/conv2 pnp_utils.py:78,107; Upsample2D/Downsample2D pnp_utils.py:51-76);
 *                       asym = 1: zero padding only after the last row / column (F.pad (0,1,0,1) + stride-2
 *                       conv of the AutoencoderKL encoder's Downsample2D, pipeline_i2vgen_xl.py:565-592)
 *   mode 2  TEMPORAL3 : 3 taps along the frame axis, pad 1 (Conv3d (3,1,1) of TemporalConvLayer;
 *                       consisti2v/consisti2v/models/videoldm_unet_blocks.py:316-328)
 * epilogue: + bias[n]; + rowvec[(m / rowvec_div) * ldrv + n] (the temb broadcast pnp_utils.py:91-92);
 *           act: 0 none, 1 SiLU, 2 GELU(erf), 3 GEGLU (W packed [16 x h | 16 x gate] per 32 rows, out N/2),
 *                4 = store the fp32 accumulator (+bias) to C as float (ldc in floats; no rowvec / R): attention
 *                logits of the VAE mid-block's single 512-wide head;
 *           + R[m, n] (residual, pnp_utils.py:124); store fp16.
 */
typedef struct AnyV2VGemmDesc {
    const void* A0;
    const void* A1;      /* may be NULL when C1 == 0 */
    const void* W;       /* [N][taps*(C0+C1)] */
    void* C;
    const void* bias;    /* [N] or NULL */
    const void* rowvec;  /* [M/rowvec_div][ldrv] or NULL */
    const void* R;       /* [M][ldr] or NULL */
    int32_t M, N;
    int32_t C0, C1;
    int32_t lda0, lda1, ldc, ldr, ldrv, rowvec_div;
    int32_t mode;
    int32_t Hi, Wi, Ho, Wo, stride, up; /* mode 1 */
    int32_t asym;                       /* mode 1: 0 = pad 1 on every side, 1 = pad only right / bottom */
    int32_t F, HW;                      /* mode 2: frames per clip, pixels per frame */
    int32_t act;
    int32_t flags;       /* bit0: force the naive reference kernel; bit1: LDS-DMA staging; bit2: never use the
                            persistent 192x320 kernel (nor the weight-stationary one); bit3: always use it when the shape
                            allows; bit4: no split-K; bit9 (512): never use the weight-stationary K = 320 kernel; bit10 (1024):
                            use it whenever the shape allows (mode 0, C0 = 320 with N % 160 = 0 or C0 = 512 with N % 64 = 0 (GEGLU: 128), C1 = 0, act 0 | 3, no rowvec), also
                            below its M >= 32768 threshold; bit11 (2048) / bit12 (4096): 128-column / 160-column tiles in the
                            128-row kernel regardless of the fill heuristic; bits 13-15: tile order of the persistent kernel on
                            wide-N launches (0 auto, 1 classic N-fastest, 2..6 super-tiles of 4 / 8 / 16 / 32 / 2 M-tiles per XCD
                            round), bit16: super-tiles walked N-fastest -- every order gives bit-identical results; bit17: take the
                            ping-pong persistent kernel wherever the shape allows (N % 320 = 0, no GEGLU), bit18: never take it,
                            bit19 / bit20: its 192- / 256-row tile (bit-identical to the other tile kernels); round 6, all OFF by
                            default (A/B switches and tests): bit21: take the one-wave-per-SIMD persistent kernel (gemm_sw.hip:
                            N % 320 = 0, >= 2 K-tiles; bit-identical to the other tile kernels), bit22: never take any of the round-6
                            kernels, bit26 / bit27: allow / force its stream-K form (un-hinted launches, needs a workspace of 126 MB;
                            a different fp32 summation order), bit28: 3x3 stride-1 "same" convolutions at image width 16 / 32 / 64 on
                            the LDS-patch kernel (gemm_swh.hip: K order (dy, slice, dx), a different summation order).  All other
                            bits are ignored by the product library. */
    void* workspace;     /* optional fp32 scratch for split-K partial tiles (small-M, long-K launches) or NULL */
    int64_t workspace_bytes;
    /* LayerNorm folded into the projection that consumes it (BasicTransformerBlock.norm1/2/3 -> attn.to_q/k/v / ff.net[0].proj,
     * consisti2v/consisti2v/models/videoldm_transformer_blocks.py:461-564): with ln_c1 != NULL the rows of A0 are the
     * UN-normalised residual stream and
     *     C = act( rstd[m] * (A0 W^T - mean[m] * ln_c1) + bias ),   mean / rstd = LayerNorm statistics of row m over C0, eps = ln_eps,
     * where the caller passes W = W_proj diag(gamma) (fp16), ln_c1[n] = sum_k W[n][k] (fp32, summed over the fp16-rounded W) and
     * bias = b_proj + W_proj beta.  Mode 0, C0 = 320 (N % 160 = 0) or C0 = 512 with GEGLU (N % 128 = 0), no rowvec / R: other
     * shapes return ANYV2V_EUNSUPPORTED (the caller then runs anyv2v_layernorm_f16 + the plain GEMM). */
    const float* ln_c1;
    float ln_eps;
    int32_t reserved0;
} AnyV2VGemmDesc;

int anyv2v_gemm_f16(const AnyV2VGemmDesc* d, void* stream);

/* ---- fused feed-forward ----------------------------------------------------------------------
 * Y = GEGLU(X W1^T + b1) W2^T + b2 (+ R): the FeedForward of BasicTransformerBlock (diffusers-0.26.3 `FeedForward(dim,
 * activation_fn="geglu")` behind pipeline_i2vgen_xl.py:1146; in-tree restatement consisti2v/consisti2v/models/
 * videoldm_transformer_blocks.py:545-563) in one kernel -- the [M, H] hidden activation never reaches HBM.  Implemented for
 * C = 320, H = 1280 (the 64x64 level); anything else returns ANYV2V_EUNSUPPORTED and the caller runs the two GEMMs.
 *   W1 [2 H][C], b1 [2 H]: rows interleaved [16 x h | 16 x gate] per 32 (the act = 3 packing of anyv2v_gemm_f16);
 *   W2 [H / 32][C][32]   : slab-major; inside a slab of 32 hidden units, column 8 q + e (q = 0..3) holds hidden unit 4 q + e for
 *                          e < 4 and 16 + 4 q + (e - 4) for e >= 4 (the MFMA K-slot order the kernel produces the hidden values in);
 *   b2 [C]; R [M][ldr] or NULL; X [M][ldx]; Y [M][ldy].  fp32 accumulation, the hidden value rounded to fp16 once, Y = fp16(acc + b2)
 *   then + R in fp16 (the rounding points of the unfused GEGLU GEMM + Linear pair). */
typedef struct AnyV2VFFDesc {
    const void* X;
    const void* W1;
    const void* b1;
    const void* W2;
    const void* b2;
    const void* R;
    void* Y;
    int32_t M, C, H;
    int32_t ldx, ldr, ldy;
    int32_t flags;       /* reserved, 0 */
    int32_t reserved0;
} AnyV2VFFDesc;

int anyv2v_ff_geglu_f16(const AnyV2VFFDesc* d, void* stream);

/* ---- normalisation ---------------------------------------------------------------------------
 * GroupNorm over channels-last tokens, optionally fused SiLU, input = channel concat [X0 | X1].
 * Statistics are taken over `rows_per_group` consecutive rows x (C/G) channels: rows_per_group = H*W
 * gives the 4-D per-frame GroupNorm (ResnetBlock2D.norm1/norm2 pnp_utils.py:48,104; Transformer2DModel.norm),
 * rows_per_group = F*H*W the 5-D per-clip one (TemporalConvLayer, TransformerTemporalModel.norm).
 * `stats` is caller-provided scratch of anyv2v_groupnorm_scratch_floats(M, rows_per_group, G) floats (mean/rstd
 * plus per-chunk partial sums; the reduction uses no atomics, so results are bit-reproducible).
 */
int64_t anyv2v_groupnorm_scratch_floats(int32_t M, int32_t rows_per_group, int32_t G);
int anyv2v_groupnorm_f16(const void* X0, const void* X1, int32_t C0, int32_t C1, void* Y, const void* gamma,
                         const void* beta, float* stats, int32_t M, int32_t rows_per_group, int32_t G, float eps,
                         int32_t silu, void* stream);

/* Sharded 5-D GroupNorm, for a clip whose frames / pixels are split over `shards` ranks (the 128-frame mode,
 * gradio_demo.py:129-131; every rank holds the same local shape).  Phase 1 writes this rank's partial sums into
 * `stats`; the caller adds the first anyv2v_groupnorm_partial_floats(M, rows_per_group, G, C0 + C1) floats over the
 * ranks (one all-reduce SUM: RCCL on the node); phase 2 normalises with shards x the local element count.
 * partial + apply(shards = 1) on one rank is exactly anyv2v_groupnorm_f16. */
int64_t anyv2v_groupnorm_partial_floats(int32_t M, int32_t rows_per_group, int32_t G, int32_t C);
int anyv2v_groupnorm_partial_f16(const void* X0, const void* X1, int32_t C0, int32_t C1, float* stats, int32_t M,
                                 int32_t rows_per_group, int32_t G, void* stream);
int anyv2v_groupnorm_apply_f16(const void* X0, const void* X1, int32_t C0, int32_t C1, void* Y, const void* gamma,
                               const void* beta, const float* stats, int32_t M, int32_t rows_per_group, int32_t G,
                               float eps, int32_t silu, int32_t shards, void* stream);

/* LayerNorm over the last dim (BasicTransformerBlock.norm1/2/3). */
int anyv2v_layernorm_f16(const void* X, void* Y, const void* gamma, const void* beta, int32_t M, int32_t C,
                         float eps, void* stream);

/* ---- attention -------------------------------------------------------------------------------
 * softmax(Q K^T / sqrt(64)) V for head_dim 64, no mask -- F.scaled_dot_product_attention at
 * i2vgen-xl/pnp_utils.py:208-210 (spatial) and :314-316 (temporal).
 * Batch element i (0 <= i < batch), sequence position s, head h address row
 *     row = (i / inner) * outer_stride + (i % inner) * inner_stride + s * seq_stride
 * of a token matrix with leading dim ld; head h occupies columns [64 h, 64 h + 64).
 *   spatial : inner = 1,  outer_stride = S,    seq_stride = 1
 *   temporal: inner = HW, outer_stride = F*HW, inner_stride = 1, seq_stride = HW   (no permute needed)
 * K/V use batch index i / kv_div (cross-attention: all F frames of a clip share one K/V).
 * PnP injection (pnp_utils.py:189-196, :295-302): qk_mod > 0 makes Q and K of batch element i come
 * from element i % qk_mod (the source branch) -- aliasing instead of the reference's copies.
 */
typedef struct AnyV2VAttnDesc {
    const void* Q;
    const void* K;
    const void* V;
    void* O;
    int32_t ldq, ldk, ldv, ldo;
    int32_t batch, heads, Sq, Sk;
    int32_t inner;
    int64_t q_outer, q_inner, q_seq;     /* strides in rows */
    int64_t kv_outer, kv_inner, kv_seq;
    int32_t kv_div;
    int32_t qk_mod;
    float scale;
    int32_t flags;      /* bit0: force the naive reference kernel; bit1: no short-sequence kernel; bit2: never use the
                           8-wave (256-query) blocks; bit3: PnP launches (batch == 3 qk_mod) as per-branch aliasing on the
                           plain kernel instead of the shared-softmax kernel; bit7 (128): 8-byte instead of 16-byte epilogue stores in
                           the flash kernels (A/B of the widened epilogue; same bytes, same results).  All other bits are ignored. */
} AnyV2VAttnDesc;

int anyv2v_attention_f16(const AnyV2VAttnDesc* d, void* stream);

/* Attention for any head_dim <= 160, used once per clip (and by the ConsistI2V hook family's temporal attention, head_dim = C / 8): image_latents_temporal_encoder (2 heads x dim 4) and the CLIP towers of
 * encode_prompt / _encode_image (pipeline_i2vgen_xl.py:224-441: text 16 heads x 64 with the causal mask, vision 16 heads x 80).
 * head_dim a multiple of 8 in 40..128 and Sk <= 288, or in 136..160 and Sk <= 96: a whole-sequence MFMA kernel (K and V^T of a
 * head in LDS, exact softmax over the score row block in registers); anything else, or flags bit0: one thread per (batch, head, query).  Same addressing as above
 * with explicit head_dim; flags bit4 (16): causal mask (key j visible to query s iff j <= s). */
int anyv2v_attention_small_f16(const AnyV2VAttnDesc* d, int32_t head_dim, void* stream);
/* The same with an additive score bias: softmax(scale * Q K^T + bias[head]) V, bias fp32 [heads, Sq, Sk] shared by all batch
 * elements -- SEINE's TemporalAttention adds a learned relative-position bias to the temporal scores
 * (seine/models/attention.py:815-817,870-889; seine/pnp_utils.py:386-451 is its hooked form).  Generic kernel, head_dim <= 160. */
int anyv2v_attention_bias_f16(const AnyV2VAttnDesc* d, int32_t head_dim, const float* bias, void* stream);

/* ---- elementwise / layout --------------------------------------------------------------------- */
/* y = silu(x) (n elements) */
int anyv2v_silu_f16(const void* X, void* Y, int64_t n, void* stream);
/* y = a + b */
int anyv2v_add_f16(const void* A, const void* B, void* Y, int64_t n, void* stream);
/* sinusoidal timestep embedding, flip_sin_to_cos, freq shift 0: out[b, :] = [cos(t_b w) | sin(t_b w)] */
int anyv2v_timestep_embedding_f16(const float* t, void* out, int32_t B, int32_t dim, void* stream);
/* [B, C, F, H, W] (NCFHW, the pipeline's latent layout) -> tokens [(B F) (H W), ldy] at column col0 */
int anyv2v_ncfhw_to_tokens_f16(const void* X, void* Y, int32_t B, int32_t C, int32_t F, int32_t HW, int32_t ldy,
                               int32_t col0, void* stream);
int anyv2v_tokens_to_ncfhw_f16(const void* X, void* Y, int32_t B, int32_t C, int32_t F, int32_t HW, int32_t ldx,
                               int32_t col0, void* stream);
/* AdaptiveAvgPool2d over channels-last tokens [N, Hi, Wi, C] -> [N, Ho, Wo, C] */
int anyv2v_adaptive_avgpool_f16(const void* X, void* Y, int32_t N, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                int32_t C, void* stream);
/* Row gather with column windows: Y[m, ycol0 : ycol0+C] = X[idx[m], xcol0 : xcol0+C] (idx: int32 on the device; C, ld, col0
 * multiples of 8).  Builds the key / value sequences of ConsistI2V's first-frame-conditioned attention without a second
 * projection: spatial attn1 attends over [own frame ; first frame] (videoldm_transformer_blocks.py:479-489), temporal attn1 over
 * [the pixel's frames ; the 8 neighbours of the pixel in the first frame] (:490-503, videoldm_attention.py:589-599) -- both are
 * rows of the K / V projections that already exist. */
int anyv2v_gather_rows_f16(const void* X, int32_t ldx, int32_t xcol0, const int32_t* idx, void* Y, int32_t ldy, int32_t ycol0,
                           int64_t M, int32_t C, void* stream);
/* Rotary position embedding in place (consisti2v/consisti2v/models/rotary_embedding.py:29-49,143-163 as called by
 * RotaryEmbAttnProcessor2_0 / ModifiedTmpAttnProcessor, videoldm_attention.py:773-777, consisti2v/pnp_utils.py:306-310):
 * columns [col0 + w * window_stride, ... + rot_dim) of row r for w < n_windows, in interleaved pairs (2i, 2i+1), rotated by
 * pos(r) * theta^(-2i/rot_dim) with pos(r) = (r / rows_per_pos) % n_pos -- for token matrices [(b f)(h w), C]: rows_per_pos = HW,
 * n_pos = F.  ConsistI2V rotates ONE window (the first half of the channels, before the head split); SEINE rotates the first 32
 * channels of EVERY head (seine/models/attention.py:880-882, RotaryEmbedding(32) on [b, heads, f, d]): n_windows = heads,
 * window_stride = head_dim.  fp32 angles. */
int anyv2v_rotary_f16(void* X, int32_t ld, int64_t rows, int32_t col0, int32_t rot_dim, int32_t n_windows, int32_t window_stride,
                      int32_t rows_per_pos, int32_t n_pos, float theta, void* stream);
/* rows copy with column window: Y[m, ycol0 : ycol0+C] = X[m, xcol0 : xcol0+C] */
int anyv2v_copy_cols_f16(const void* X, int32_t ldx, int32_t xcol0, void* Y, int32_t ldy, int32_t ycol0, int64_t M,
                         int32_t C, void* stream);

/* Row softmax of fp32 logits (from anyv2v_gemm_f16 with act = 4) into fp16 probabilities:
 * P[r, c] = softmax_c(scale * S[r, c]), cols <= 8192.  AutoencoderKL mid-block attention (single head of width 512,
 * encode_vae_video / decode_latents, pipeline_i2vgen_xl.py:565-592,598-620). */
int anyv2v_softmax_rows_f32_f16(const float* S, int32_t lds, void* P, int32_t ldp, int32_t rows, int32_t cols, float scale,
                                void* stream);

/* Fused classifier-free-guidance combine + DDIM step (eta = 0, v-prediction), reading the UNet's
 * channels-last v-prediction tokens directly and updating NCFHW latents in place:
 *   v = v_unc + g * (v_cond - v_unc)                              pipeline_i2vgen_xl.py:1160-1162
 *   x0 = sa_t x - sb_t v ; eps = sa_t v + sb_t x ; x' = sa_p x0 + sb_p eps   (DDIMScheduler.step, :1173;
 *   inverse: consisti2v/ddim_inverse_scheduler.py:329-369 -- same formula, different alphas)
 * Vtok: [(nb F) HW, ldv] tokens; branch `b_unc` / `b_cond` select the batch slices (b_unc < 0: no CFG).
 * coef: 4 floats on the device {sa_t, sb_t, sa_p, sb_p}.  lat/out: [1, C, F, H, W] fp16; out may alias lat.
 */
int anyv2v_cfg_ddim_step_f16(const void* Vtok, int32_t ldv, int32_t b_unc, int32_t b_cond, float guidance,
                             const float* coef, const void* lat, void* out, int32_t C, int32_t F, int32_t HW,
                             void* stream);

/* Plain elementwise DDIM / inverse-DDIM step (eta = 0, v-prediction) on same-layout tensors:
 * the generic `scheduler.step(model_output, t, sample).prev_sample` (pipeline_i2vgen_xl.py:868,1173,1418). */
int anyv2v_ddim_step_f16(const void* V, const void* X, void* Y, float sa_t, float sb_t, float sa_p, float sb_p,
                         int64_t n, void* stream);

/* Guidance combine + DDIM / inverse-DDIM step (eta = 0) on same-layout tensors, any prediction type: the loop body of the
 * ConsistI2V pipeline (consisti2v/consisti2v/pipelines/pipeline_video_editing.py:921-937 invert, :1539-1555 sample_with_pnp, :676-692
 * __call__; scheduler arithmetic consisti2v/ddim_inverse_scheduler.py:329-369).  E: the UNet's prediction of every branch, `n`
 * contiguous elements per branch; branch indices b_unc / b_img / b_txt:
 *   e = e[b_unc] + g_img (e[b_img] - e[b_unc]) + g_txt (e[b_txt] - e[b_img])    "both";   b_img < 0: e[b_unc] + g_txt (e[b_txt] - e[b_unc]);
 *   b_unc < 0: e[b_txt] (no guidance).   prediction: 0 v_prediction, 1 epsilon, 2 sample.   X / Y: `n` elements; Y may alias X. */
int anyv2v_guided_step_f16(const void* E, int64_t n, int32_t b_unc, int32_t b_img, int32_t b_txt, float g_img, float g_txt,
                           int32_t prediction, float sa_t, float sb_t, float sa_p, float sb_p, const void* X, void* Y, void* stream);

/* The same with general coefficients and an additive noise term -- the ancestral (DDPM) step of SEINE's edit loop
 * (seine/run_pnp_edit.py:205 `self.scheduler.step(noise_pred, t, x)` with sample_method "ddpm"):
 *   y = c_x0 * x0 + c_eps * eps + sigma * noise[i]      x0 / eps from the guided prediction as above (sa_t, sb_t).
 * DDPM without sample clipping: c_x0 = coeff_x0 + coeff_xt * sa_t, c_eps = coeff_xt * sb_t (x = sa_t x0 + sb_t eps).  noise may be NULL
 * when sigma is 0 (the last step). */
int anyv2v_guided_step_noise_f16(const void* E, int64_t n, int32_t b_unc, int32_t b_img, int32_t b_txt, float g_img, float g_txt,
                                 int32_t prediction, float sa_t, float sb_t, float c_x0, float c_eps, const void* X, void* Y,
                                 const void* noise, float sigma, void* stream);

/* ---- misc ------------------------------------------------------------------------------------- */
/* Launch heuristics (kernel family, split-K factor, GroupNorm chunking) see rows * num / den from now on; grids and bounds keep the true
 * row counts.  The PnP edit runs some steps on [negative, editing] only (steps outside every injection schedule; steps whose source
 * features are replayed from a multi-edit cache, pipeline_i2vgen_xl.py:1136-1162): with the hint 3 / 2 those launches choose what the
 * three-branch launch chooses, so every fp32 summation order -- and the result, bit for bit -- is the same.  (1, 1) resets.
 * PROCESS-GLOBAL and NOT thread-safe: one host-side pair of integers read by every launch at enqueue (= graph capture) time, like the
 * single-threaded reference loop it serves.  A host that enqueues from several threads must serialise "set hint .. launches ..
 * reset" itself (anyv2v_last_error(), by contrast, is thread-local). */
int anyv2v_set_batch_hint(int32_t num, int32_t den);
const char* anyv2v_last_error(void);
/* ABI version = major * 100 + minor.  Descriptors carry no size field: a caller MUST be compiled against the header of the
 * library it loads (check anyv2v_version() >= the ANYV2V_ABI_VERSION it was built with) and MUST zero-initialise every
 * descriptor (new fields are appended with 0 = "off").  101: AnyV2VGemmDesc grew ln_c1 / ln_eps / reserved0 (round 3), flags
 * bits 13-16 select the persistent kernel's tile order (round 4).  102: anyv2v_ff_geglu_f16.  103: anyv2v_guided_step_f16, anyv2v_guided_step_noise_f16. */
#define ANYV2V_ABI_VERSION 103
int anyv2v_version(void);
/* MFMA / LDS layout self-test used by the gpu test-suite (returns 0 when the layouts the kernels assume hold) */
int anyv2v_selftest(void* scratch, int64_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ANYV2V_HIP_H */
