// GroupNorm (+SiLU) and LayerNorm over channels-last token matrices (HBM-bound kernels, 16-byte vector access).
//
// GroupNorm replaces diffusers GroupNorm(32) as used by ResnetBlock2D.norm1/norm2 (reference restatement
// i2vgen-xl/pnp_utils.py:48,104), Transformer2DModel.norm, TemporalConvLayer / TransformerTemporalModel.norm
// (5-D: statistics over all frames of a clip) and conv_norm_out.  The input may be the channel concat [X0 | X1]
// (skip connection of the up blocks, consisti2v/.../videoldm_unet_blocks.py:721-745) without materialising it.
// Two kernels, no atomics (bit-reproducible): (1) partial sums per (stat group, row chunk, channel group) reduced
// through LDS; (2) normalise + affine (+ SiLU) -> fp16, each block first folding the chunk sums of its stat group into
// (mean, rstd) in a fixed order (a separate 7-us finalize launch per GroupNorm used to cost 2 % of the step; a
// last-arriver finalize would need device-scope fences, which flush the per-XCD L2 on gfx950).
// Algorithmic traffic: 2 reads + 1 write of X.
#include "common.h"

#define GN_MAX_CHUNKS 256

// (1) per-(stat group, row chunk) partial sums, reduced deterministically through LDS (no atomics)
__global__ void gn_partial_kernel(const half_t* __restrict__ X0, const half_t* __restrict__ X1, int C0, int C1,
                                  float* __restrict__ partial, int rows_per_group, int G, int rows_chunk, int rpb,
                                  int nchunks) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [rpb][C][2]
    const int C = C0 + C1, V = C >> 3, cpg = C / G;
    const int sg = blockIdx.x, chunk = blockIdx.y;
    const int tid = threadIdx.x;
    const int rl = tid / V, v = tid - rl * V;
    if (rl < rpb) {
        const int r_begin = chunk * rows_chunk;
        int r_end = r_begin + rows_chunk;
        if (r_end > rows_per_group) r_end = rows_per_group;
        const int c0 = v * 8;
        const bool from0 = c0 < C0;
        const half_t* base = from0 ? X0 : X1;
        const int ld = from0 ? C0 : C1;
        const int cc = from0 ? c0 : c0 - C0;
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
        const half_t* ptr = base + ((size_t)sg * rows_per_group) * ld + cc;
        int r = r_begin + rl;
        for (; r + 3 * rpb < r_end; r += 4 * rpb) {  // 4 independent 16-byte loads in flight
            h8 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = *(const h8*)(ptr + (size_t)(r + u * rpb) * ld);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)x[u][e];
                    s[e] += f;
                    q[e] += f * f;
                }
        }
        for (; r < r_end; r += rpb) {
            const h8 x = *(const h8*)(ptr + (size_t)r * ld);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)x[e];
                s[e] += f;
                q[e] += f * f;
            }
        }
        float* dst = red + ((size_t)rl * C + c0) * 2;
#pragma unroll
        for (int e = 0; e < 8; e += 2) *(f4*)(dst + 2 * e) = (f4){s[e], q[e], s[e + 1], q[e + 1]};
    }
    __syncthreads();
    if (tid < G) {
        float as = 0.f, aq = 0.f;
        for (int rr = 0; rr < rpb; ++rr) {
            const float* src = red + ((size_t)rr * C + tid * cpg) * 2;
            for (int c = 0; c < cpg; ++c) {
                as += src[2 * c];
                aq += src[2 * c + 1];
            }
        }
        float* o = partial + (((size_t)sg * nchunks + chunk) * G + tid) * 2;
        o[0] = as;
        o[1] = aq;
    }
}

// (2) normalise + affine (+ SiLU); grid = (row blocks per stat group, stat groups), block = V x rpb threads.
// A thread owns ONE 16-byte channel vector (its 8 channels' mean / rstd / gamma / beta live in registers for the whole
// kernel: no index division, no per-element group stepping, no gamma / beta reloads) and walks the rows of its block
// with U row loads in flight (the previous one-vector-per-iteration grid-stride form had 32 KiB in flight per CU and
// stopped at 3.6 TB/s read + write).
__device__ __forceinline__ float gn_silu(float x) {  // x * sigmoid(x) on raw v_exp / v_rcp (1 ulp; the output is fp16)
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

template <int U>
__global__ __launch_bounds__(1024) void gn_apply_kernel(const half_t* __restrict__ X0, const half_t* __restrict__ X1, int C0,
                                                        int C1, half_t* __restrict__ Y, const half_t* __restrict__ gamma,
                                                        const half_t* __restrict__ beta, const float* __restrict__ partial,
                                                        int nchunks, float inv_cnt, float eps, int rows_per_group, int G,
                                                        int silu, int rpb, int rows_per_block) {
    __shared__ float rs[256], rq[256], smean[64], srstd[64];
    const int C = C0 + C1, V = C >> 3, cpg = C / G;
    const int sg = blockIdx.y, tid = threadIdx.x;
    {   // statistics of this stat group: same order of additions in every block -> identical in all of them
        const int nt = blockDim.x < 256 ? blockDim.x : 256;
        const int parts = nt / G;  // G <= 64 <= blockDim.x
        const int g = tid % G, part = tid / G;
        if (tid < 256) {
            float as = 0.f, aq = 0.f;
            if (part < parts)
                for (int c = part; c < nchunks; c += parts) {
                    const float* src = partial + (((size_t)sg * nchunks + c) * G + g) * 2;
                    as += src[0];
                    aq += src[1];
                }
            rs[tid] = as;
            rq[tid] = aq;
        }
        __syncthreads();
        if (tid < G) {
            float s = 0.f, q = 0.f;
            for (int p2 = 0; p2 < parts; ++p2) {
                s += rs[p2 * G + tid];
                q += rq[p2 * G + tid];
            }
            const float mean = s * inv_cnt;
            const float var = fmaxf(q * inv_cnt - mean * mean, 0.f);
            smean[tid] = mean;
            srstd[tid] = rsqrtf(var + eps);
        }
        __syncthreads();
    }
    const int rl = tid / V, v = tid - rl * V;
    if (rl >= rpb) return;
    const int c0 = v * 8;
    const bool from0 = c0 < C0;
    const int ld = from0 ? C0 : C1;
    const half_t* src = (from0 ? X0 + c0 : X1 + (c0 - C0)) + (size_t)sg * rows_per_group * ld;
    half_t* dst = Y + (size_t)sg * rows_per_group * C + c0;
    float mean[8], rstd[8], ga[8], be[8];
    {
        const h8 gv = *(const h8*)(gamma + c0);
        const h8 bv = *(const h8*)(beta + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ge = (c0 + e) / cpg;
            mean[e] = smean[ge];
            rstd[e] = srstd[ge];
            ga[e] = (float)gv[e];
            be[e] = (float)bv[e];
        }
    }
    const int r_begin = blockIdx.x * rows_per_block;
    int r_end = r_begin + rows_per_block;
    if (r_end > rows_per_group) r_end = rows_per_group;
    auto norm8 = [&](const h8& x) {
        h8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = ((float)x[e] - mean[e]) * rstd[e] * ga[e] + be[e];
            if (silu) f = gn_silu(f);
            y[e] = (half_t)f;
        }
        return y;
    };
    int r = r_begin + rl;
    for (; r + (U - 1) * rpb < r_end; r += U * rpb) {
        h8 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = *(const h8*)(src + (size_t)(r + u * rpb) * ld);
#pragma unroll
        for (int u = 0; u < U; ++u) *(h8*)(dst + (size_t)(r + u * rpb) * C) = norm8(x[u]);
    }
    for (; r < r_end; r += rpb) *(h8*)(dst + (size_t)r * C) = norm8(*(const h8*)(src + (size_t)r * ld));
}

extern "C" int64_t anyv2v_groupnorm_scratch_floats(int32_t M, int32_t rows_per_group, int32_t G) {
    const int64_t nsg = rows_per_group > 0 ? M / rows_per_group : 0;
    return nsg * G * 2 * (int64_t)(1 + GN_MAX_CHUNKS);
}

// launch plan shared by the one-call and the two-phase (sharded) entry points
struct GnPlan {
    int nsg, V, rpb, threads, nchunks, rows_chunk, rows_block;
    size_t lds;
    long long bps;
};

template <int U>
__global__ void gn_apply_kernel(const half_t*, const half_t*, int, int, half_t*, const half_t*, const half_t*, const float*, int, float,
                                float, int, int, int, int, int);

// resident gn_apply blocks on the whole device for a block size (occupancy query, cached per size)
static int gn_apply_slots(int threads) {
    static int cache[1025];
    if (threads < 1 || threads > 1024) return 2048;
    if (cache[threads] == 0) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gn_apply_kernel<4>, threads, 0) != hipSuccess || per_cu < 1) per_cu = 7;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus < 1)
            cus = 256;
        cache[threads] = per_cu * cus;
    }
    return cache[threads];
}

static int gn_plan(GnPlan& pl, const void* X0, const void* X1, int32_t C0, int32_t C1, int32_t M, int32_t rows_per_group,
                   int32_t G) {
    AV_CHECK(C0 > 0 && C1 >= 0 && (C1 == 0 || X1), "groupnorm: bad C0/C1");
    const int C = C0 + C1;
    AV_CHECK(C0 % 8 == 0 && C1 % 8 == 0, "groupnorm: C0/C1 must be multiples of 8 (%d,%d)", C0, C1);
    AV_CHECK(G > 0 && G <= 64 && C % G == 0, "groupnorm: bad group count %d for C=%d", G, C);
    AV_CHECK(rows_per_group > 0 && M % rows_per_group == 0, "groupnorm: M %% rows_per_group != 0");
    AV_CHECK(av_aligned16(X0) && av_aligned16(X1), "groupnorm: pointers must be 16-byte aligned");
    pl.nsg = M / rows_per_group;
    pl.V = C / 8;
    AV_CHECK(pl.V <= 1024, "groupnorm: C too large (%d)", C);
    pl.rpb = 256 / pl.V;
    if (pl.rpb < 1) pl.rpb = 1;
    pl.threads = pl.V * pl.rpb;
    if (pl.threads < 64) pl.threads = 64;
    if (pl.threads < G) pl.threads = G;
    // (chunking decided on the hinted number of statistics groups -- anyv2v_set_batch_hint: a two-branch step sums its chunks in
    //  the order the three-branch step does)
    const int nsg_h = av_hint_rows(M) / rows_per_group > 0 ? av_hint_rows(M) / rows_per_group : 1;
    int nchunks = (1536 + nsg_h - 1) / nsg_h;
    int max_chunks = (rows_per_group + pl.rpb * 8 - 1) / (pl.rpb * 8);  // >= 8 row-iterations per thread
    if (max_chunks < 1) max_chunks = 1;
    if (nchunks > max_chunks) nchunks = max_chunks;
    if (nchunks > GN_MAX_CHUNKS) nchunks = GN_MAX_CHUNKS;
    pl.rows_chunk = (rows_per_group + nchunks - 1) / nchunks;
    pl.nchunks = (rows_per_group + pl.rows_chunk - 1) / pl.rows_chunk;
    pl.lds = (size_t)pl.rpb * C * 2 * sizeof(float);
    AV_CHECK(pl.lds <= 64 * 1024, "groupnorm: LDS reduction buffer too large");
    // Apply blocks: as many as the chip holds at once (CUs x resident blocks of this size), not more -- 2064 blocks on 1792 slots
    // ran a second, 15 %-full round (the 64x64-level launches, profiles/r03_bench_kernel_summary.md); each block at least 8
    // row-iterations per thread where the stat group is large enough.  (The partition has no effect on the result.)
    const long long vec_sg = (long long)rows_per_group * pl.V;
    AV_CHECK(vec_sg < (1ll << 31), "groupnorm: stat group too large (%lld vectors)", vec_sg);
    const int slots = gn_apply_slots(pl.threads);
    pl.bps = slots / pl.nsg;
    const long long max_bps = (rows_per_group + pl.rpb * 8 - 1) / (pl.rpb * 8);
    if (pl.bps > max_bps) pl.bps = max_bps;
    if (pl.bps < 1) pl.bps = 1;
    pl.rows_block = (int)((rows_per_group + pl.bps - 1) / pl.bps);
    pl.bps = (rows_per_group + pl.rows_block - 1) / pl.rows_block;
    return ANYV2V_OK;
}

static int gn_partial(const GnPlan& pl, const void* X0, const void* X1, int32_t C0, int32_t C1, float* stats,
                      int32_t rows_per_group, int32_t G, hipStream_t s) {
    // stats: [nsg][nchunks][G][2]
    hipLaunchKernelGGL(gn_partial_kernel, dim3(pl.nsg, pl.nchunks), dim3(pl.threads), pl.lds, s, (const half_t*)X0,
                       (const half_t*)X1, C0, C1, stats, rows_per_group, G, pl.rows_chunk, pl.rpb, pl.nchunks);
    return av_launch_status("groupnorm<partial>");
}

static int gn_apply(const GnPlan& pl, const void* X0, const void* X1, int32_t C0, int32_t C1, void* Y, const void* gamma,
                    const void* beta, const float* stats, int32_t rows_per_group, int32_t G, float eps, int32_t silu,
                    int32_t shards, hipStream_t s) {
    AV_CHECK(av_aligned16(Y) && av_aligned16(gamma) && av_aligned16(beta), "groupnorm: pointers must be 16-byte aligned");
    const float inv_cnt = 1.0f / ((float)rows_per_group * (float)((C0 + C1) / G) * (float)shards);
    hipLaunchKernelGGL(gn_apply_kernel<4>, dim3((unsigned)pl.bps, (unsigned)pl.nsg), dim3(pl.threads), 0, s,
                       (const half_t*)X0, (const half_t*)X1, C0, C1, (half_t*)Y, (const half_t*)gamma, (const half_t*)beta,
                       stats, pl.nchunks, inv_cnt, eps, rows_per_group, G, silu, pl.rpb, pl.rows_block);
    return av_launch_status("groupnorm");
}

extern "C" int anyv2v_groupnorm_f16(const void* X0, const void* X1, int32_t C0, int32_t C1, void* Y,
                                    const void* gamma, const void* beta, float* stats, int32_t M,
                                    int32_t rows_per_group, int32_t G, float eps, int32_t silu, void* stream) {
    AV_CHECK(X0 && Y && gamma && beta && stats, "groupnorm: null pointer");
    GnPlan pl;
    if (int rc = gn_plan(pl, X0, X1, C0, C1, M, rows_per_group, G)) return rc;
    if (int rc = gn_partial(pl, X0, X1, C0, C1, stats, rows_per_group, G, (hipStream_t)stream)) return rc;
    return gn_apply(pl, X0, X1, C0, C1, Y, gamma, beta, stats, rows_per_group, G, eps, silu, 1, (hipStream_t)stream);
}

// Sharded GroupNorm (a clip whose frames or pixels are split over `shards` ranks, every rank holding the same local
// shape): phase 1 writes this rank's partial sums; the caller adds the first anyv2v_groupnorm_partial_floats() floats
// of `stats` over the ranks (all-reduce SUM, RCCL); phase 2 normalises with shards x the local element count.
extern "C" int64_t anyv2v_groupnorm_partial_floats(int32_t M, int32_t rows_per_group, int32_t G, int32_t C) {
    GnPlan pl;
    static const int dummy __attribute__((aligned(16))) = 0;
    if (gn_plan(pl, &dummy, nullptr, C, 0, M, rows_per_group, G) != ANYV2V_OK) return -1;
    return (int64_t)pl.nsg * pl.nchunks * G * 2;
}

extern "C" int anyv2v_groupnorm_partial_f16(const void* X0, const void* X1, int32_t C0, int32_t C1, float* stats, int32_t M,
                                            int32_t rows_per_group, int32_t G, void* stream) {
    AV_CHECK(X0 && stats, "groupnorm_partial: null pointer");
    GnPlan pl;
    if (int rc = gn_plan(pl, X0, X1, C0, C1, M, rows_per_group, G)) return rc;
    return gn_partial(pl, X0, X1, C0, C1, stats, rows_per_group, G, (hipStream_t)stream);
}

extern "C" int anyv2v_groupnorm_apply_f16(const void* X0, const void* X1, int32_t C0, int32_t C1, void* Y,
                                          const void* gamma, const void* beta, const float* stats, int32_t M,
                                          int32_t rows_per_group, int32_t G, float eps, int32_t silu, int32_t shards,
                                          void* stream) {
    AV_CHECK(X0 && Y && gamma && beta && stats, "groupnorm_apply: null pointer");
    AV_CHECK(shards >= 1, "groupnorm_apply: shards must be >= 1");
    GnPlan pl;
    if (int rc = gn_plan(pl, X0, X1, C0, C1, M, rows_per_group, G)) return rc;
    return gn_apply(pl, X0, X1, C0, C1, Y, gamma, beta, stats, rows_per_group, G, eps, silu, shards, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row kept in registers (two-pass mean / variance), C % 8 == 0, C <= 2048.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ X, half_t* __restrict__ Y,
                                                        const half_t* __restrict__ gamma,
                                                        const half_t* __restrict__ beta, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int V = C >> 3;
    const int stride = gridDim.x * wpb;
    // gamma / beta of this lane's vectors: loaded once per wave, not once per row
    h8 ga[NV], be[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        if (v < V) {
            ga[i] = *(const h8*)(gamma + v * 8);
            be[i] = *(const h8*)(beta + v * 8);
        }
    }
    auto load_row = [&](int row, h8 (&xv)[NV]) {
        const half_t* x = X + (size_t)row * C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < V) xv[i] = *(const h8*)(x + v * 8);
        }
    };
    auto norm_row = [&](int row, const h8 (&xv)[NV]) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < V) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += (float)xv[i][e];
            }
        const float mean = wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < V) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)xv[i][e] - mean;
                    sq += d * d;
                }
            }
        const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
        half_t* y = Y + (size_t)row * C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < V) {
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = (half_t)(((float)xv[i][e] - mean) * rstd * (float)ga[i][e] + (float)be[i][e]);
                *(h8*)(y + v * 8) = o;
            }
        }
    };
    // two rows in flight per wave (one row per wave and iteration left ~20 KiB in flight per CU at C = 320)
    int row = blockIdx.x * wpb + (threadIdx.x >> 6);
    for (; row + stride < M; row += 2 * stride) {
        h8 xa[NV], xb[NV];
        load_row(row, xa);
        load_row(row + stride, xb);
        norm_row(row, xa);
        norm_row(row + stride, xb);
    }
    if (row < M) {
        h8 xa[NV];
        load_row(row, xa);
        norm_row(row, xa);
    }
}

// any C (tiny rows, e.g. C = 4 of image_latents_temporal_encoder.norm1): one thread per row
__global__ void layernorm_naive_kernel(const half_t* __restrict__ X, half_t* __restrict__ Y,
                                       const half_t* __restrict__ gamma, const half_t* __restrict__ beta, int M, int C,
                                       float eps) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M) return;
    const half_t* x = X + (size_t)row * C;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += (float)x[c];
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = (float)x[c] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(q / (float)C + eps);
    for (int c = 0; c < C; ++c)
        Y[(size_t)row * C + c] = (half_t)(((float)x[c] - mean) * rstd * (float)gamma[c] + (float)beta[c]);
}

extern "C" int anyv2v_layernorm_f16(const void* X, void* Y, const void* gamma, const void* beta, int32_t M, int32_t C,
                                    float eps, void* stream) {
    AV_CHECK(X && Y && gamma && beta && M > 0 && C > 0, "layernorm: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const bool fast = C % 8 == 0 && C <= 2048 && av_aligned16(X) && av_aligned16(Y) && av_aligned16(gamma) &&
                      av_aligned16(beta);
    if (!fast) {
        hipLaunchKernelGGL(layernorm_naive_kernel, dim3((M + 255) / 256), dim3(256), 0, s, (const half_t*)X, (half_t*)Y,
                           (const half_t*)gamma, (const half_t*)beta, M, C, eps);
        return av_launch_status("layernorm_naive");
    }
    int blocks = (M + 7) / 8;  // 4 waves per block, two rows in flight per wave
    if (blocks > 8192) blocks = 8192;
    const int V = C / 8;
#define LN_LAUNCH(NV)                                                                                             \
    hipLaunchKernelGGL(layernorm_kernel<NV>, dim3(blocks), dim3(256), 0, s, (const half_t*)X, (half_t*)Y,         \
                       (const half_t*)gamma, (const half_t*)beta, M, C, eps)
    if (V <= 64) LN_LAUNCH(1);
    else if (V <= 128) LN_LAUNCH(2);
    else if (V <= 192) LN_LAUNCH(3);
    else LN_LAUNCH(4);
#undef LN_LAUNCH
    return av_launch_status("layernorm");
}
