// GroupNorm (+SiLU) and LayerNorm over channels-last token matrices (HBM-bound kernels, 16-byte vector access).
//
// GroupNorm replaces diffusers GroupNorm(32) as used by ResnetBlock2D.norm1/norm2 (reference restatement
// i2vgen-xl/pnp_utils.py:48,104), Transformer2DModel.norm, TemporalConvLayer / TransformerTemporalModel.norm
// (5-D: statistics over all frames of a clip) and conv_norm_out.  The input may be the channel concat [X0 | X1]
// (skip connection of the up blocks, consisti2v/.../videoldm_unet_blocks.py:721-745) without materialising it.
// Two kernels: (1) partial sums per (stat-group, channel-group) reduced through LDS then global atomics into a
// zeroed scratch, (2) normalise + affine (+ SiLU) -> fp16.  Algorithmic traffic: 2 reads + 1 write of X.
#include "common.h"

__global__ void gn_stats_kernel(const half_t* __restrict__ X0, const half_t* __restrict__ X1, int C0, int C1,
                                float* __restrict__ stats, int rows_per_group, int G, int rows_chunk, int rpb) {
    __shared__ float ssum[64], ssq[64];
    const int C = C0 + C1, V = C >> 3, cpg = C / G;
    const int sg = blockIdx.x, chunk = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < 64) {
        ssum[tid] = 0.f;
        ssq[tid] = 0.f;
    }
    __syncthreads();
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
        const size_t row0 = (size_t)sg * rows_per_group;
        for (int r = r_begin + rl; r < r_end; r += rpb) {
            const h8 x = *(const h8*)(base + (row0 + r) * ld + cc);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)x[e];
                s[e] += f;
                q[e] += f * f;
            }
        }
        int g = c0 / cpg;
        float as = 0.f, aq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ge = (c0 + e) / cpg;
            if (ge != g) {
                atomicAdd(&ssum[g], as);
                atomicAdd(&ssq[g], aq);
                as = aq = 0.f;
                g = ge;
            }
            as += s[e];
            aq += q[e];
        }
        atomicAdd(&ssum[g], as);
        atomicAdd(&ssq[g], aq);
    }
    __syncthreads();
    if (tid < G) {
        atomicAdd(&stats[((size_t)sg * G + tid) * 2 + 0], ssum[tid]);
        atomicAdd(&stats[((size_t)sg * G + tid) * 2 + 1], ssq[tid]);
    }
}

__global__ void gn_apply_kernel(const half_t* __restrict__ X0, const half_t* __restrict__ X1, int C0, int C1,
                                half_t* __restrict__ Y, const half_t* __restrict__ gamma,
                                const half_t* __restrict__ beta, const float* __restrict__ stats, long long M,
                                int rows_per_group, int G, float eps, int silu) {
    const int C = C0 + C1, V = C >> 3, cpg = C / G;
    const float inv_cnt = 1.0f / ((float)rows_per_group * (float)cpg);
    const long long total = M * V;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / V;
        const int v = (int)(idx - row * V);
        const int c0 = v * 8;
        const int sg = (int)(row / rows_per_group);
        h8 x;
        if (c0 < C0)
            x = *(const h8*)(X0 + row * C0 + c0);
        else
            x = *(const h8*)(X1 + row * C1 + (c0 - C0));
        const h8 ga = *(const h8*)(gamma + c0);
        const h8 be = *(const h8*)(beta + c0);
        h8 y;
        int g = -1;
        float mean = 0.f, rstd = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ge = (c0 + e) / cpg;
            if (ge != g) {
                g = ge;
                const float s = stats[((size_t)sg * G + g) * 2 + 0];
                const float q = stats[((size_t)sg * G + g) * 2 + 1];
                mean = s * inv_cnt;
                const float var = fmaxf(q * inv_cnt - mean * mean, 0.f);
                rstd = rsqrtf(var + eps);
            }
            float f = ((float)x[e] - mean) * rstd * (float)ga[e] + (float)be[e];
            if (silu) f = av_silu(f);
            y[e] = (half_t)f;
        }
        *(h8*)(Y + row * C + c0) = y;
    }
}

extern "C" int anyv2v_groupnorm_f16(const void* X0, const void* X1, int32_t C0, int32_t C1, void* Y,
                                    const void* gamma, const void* beta, float* stats, int32_t M,
                                    int32_t rows_per_group, int32_t G, float eps, int32_t silu, void* stream) {
    AV_CHECK(X0 && Y && gamma && beta && stats, "groupnorm: null pointer");
    AV_CHECK(C0 > 0 && C1 >= 0 && (C1 == 0 || X1), "groupnorm: bad C0/C1");
    const int C = C0 + C1;
    AV_CHECK(C0 % 8 == 0 && C1 % 8 == 0, "groupnorm: C0/C1 must be multiples of 8 (%d,%d)", C0, C1);
    AV_CHECK(G > 0 && G <= 64 && C % G == 0, "groupnorm: bad group count %d for C=%d", G, C);
    AV_CHECK(rows_per_group > 0 && M % rows_per_group == 0, "groupnorm: M %% rows_per_group != 0");
    AV_CHECK(av_aligned16(X0) && av_aligned16(X1) && av_aligned16(Y) && av_aligned16(gamma) && av_aligned16(beta),
             "groupnorm: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int nsg = M / rows_per_group;
    const int V = C / 8;
    AV_CHECK(V <= 1024, "groupnorm: C too large (%d)", C);
    hipError_t e = hipMemsetAsync(stats, 0, (size_t)nsg * G * 2 * sizeof(float), s);
    if (e != hipSuccess) {
        anyv2v_set_error("groupnorm memset: %s", hipGetErrorString(e));
        return (int)e;
    }
    int rpb = 256 / V;
    if (rpb < 1) rpb = 1;
    int threads = V * rpb;
    if (threads < 64) threads = 64;
    int nchunks = (2048 + nsg - 1) / nsg;
    int max_chunks = (rows_per_group + rpb * 4 - 1) / (rpb * 4);  // >= 4 row-iterations per thread
    if (max_chunks < 1) max_chunks = 1;
    if (nchunks > max_chunks) nchunks = max_chunks;
    if (nchunks > 65535) nchunks = 65535;
    const int rows_chunk = (rows_per_group + nchunks - 1) / nchunks;
    nchunks = (rows_per_group + rows_chunk - 1) / rows_chunk;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nsg, nchunks), dim3(threads), 0, s, (const half_t*)X0, (const half_t*)X1,
                       C0, C1, stats, rows_per_group, G, rows_chunk, rpb);
    const long long total = (long long)M * V;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const half_t*)X0, (const half_t*)X1,
                       C0, C1, (half_t*)Y, (const half_t*)gamma, (const half_t*)beta, (const float*)stats,
                       (long long)M, rows_per_group, G, eps, silu);
    return av_launch_status("groupnorm");
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
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < M; row += gridDim.x * wpb) {
        const half_t* x = X + (size_t)row * C;
        h8 xv[NV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < V) {
                xv[i] = *(const h8*)(x + v * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += (float)xv[i][e];
            }
        }
        const float mean = wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < V) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)xv[i][e] - mean;
                    sq += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
        half_t* y = Y + (size_t)row * C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < V) {
                const h8 ga = *(const h8*)(gamma + v * 8);
                const h8 be = *(const h8*)(beta + v * 8);
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = (half_t)(((float)xv[i][e] - mean) * rstd * (float)ga[e] + (float)be[e]);
                *(h8*)(y + v * 8) = o;
            }
        }
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
    int blocks = (M + 3) / 4;
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
