// HBM-bound glue kernels of the I2VGen-XL step: SiLU, add, sinusoidal timestep embedding, NCFHW <-> token
// layout, adaptive average pool, column-window copy, and the fused CFG-combine + DDIM(-inverse) step.
// Reference call sites: pipeline_i2vgen_xl.py:1160-1162 (CFG), :1168-1176 (permute + scheduler.step + permute),
// consisti2v/ddim_inverse_scheduler.py:329-369 (inverse step formula); diffusers Timesteps(flip_sin_to_cos=True).
#include "common.h"

__global__ void silu_kernel(const half_t* __restrict__ X, half_t* __restrict__ Y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        Y[i] = (half_t)av_silu((float)X[i]);
}

__global__ void add_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B, half_t* __restrict__ Y,
                           long long n8, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const h8 a = *(const h8*)(A + i * 8), b = *(const h8*)(B + i * 8);
        h8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (half_t)((float)a[e] + (float)b[e]);
        *(h8*)(Y + i * 8) = y;
    }
    for (long long i = n8 * 8 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        Y[i] = (half_t)((float)A[i] + (float)B[i]);
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, half_t* __restrict__ out, int B, int dim) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * dim) return;
    const int b = idx / dim, j = idx - b * dim;
    const int half = dim / 2;
    const int f = j < half ? j : j - half;
    const float freq = expf(-9.210340371976184f * (float)f / (float)half);
    const float arg = t[b] * freq;
    out[idx] = (half_t)(j < half ? cosf(arg) : sinf(arg));
}

__global__ void ncfhw_to_tokens_kernel(const half_t* __restrict__ X, half_t* __restrict__ Y, int B, int C, int F,
                                       int HW, int ldy, int col0) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * F * HW;
    if (idx >= total) return;
    const int p = (int)(idx % HW);
    const int f = (int)((idx / HW) % F);
    const int b = (int)(idx / ((long long)HW * F));
    for (int c = 0; c < C; ++c) Y[idx * ldy + col0 + c] = X[(((long long)b * C + c) * F + f) * HW + p];
}

__global__ void tokens_to_ncfhw_kernel(const half_t* __restrict__ X, half_t* __restrict__ Y, int B, int C, int F,
                                       int HW, int ldx, int col0) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * F * HW;
    if (idx >= total) return;
    const int p = (int)(idx % HW);
    const int f = (int)((idx / HW) % F);
    const int b = (int)(idx / ((long long)HW * F));
    for (int c = 0; c < C; ++c) Y[(((long long)b * C + c) * F + f) * HW + p] = X[idx * ldx + col0 + c];
}

__global__ void adaptive_avgpool_kernel(const half_t* __restrict__ X, half_t* __restrict__ Y, int N, int Hi, int Wi,
                                        int Ho, int Wo, int C) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * Ho * Wo * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int xo = (int)((idx / C) % Wo);
    const int yo = (int)((idx / ((long long)C * Wo)) % Ho);
    const int n = (int)(idx / ((long long)C * Wo * Ho));
    const int y0 = (yo * Hi) / Ho, y1 = ((yo + 1) * Hi + Ho - 1) / Ho;
    const int x0 = (xo * Wi) / Wo, x1 = ((xo + 1) * Wi + Wo - 1) / Wo;
    float s = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) s += (float)X[(((long long)n * Hi + y) * Wi + x) * C + c];
    Y[idx] = (half_t)(s / (float)((y1 - y0) * (x1 - x0)));
}

__global__ void copy_cols_kernel(const half_t* __restrict__ X, int ldx, int xcol0, half_t* __restrict__ Y, int ldy,
                                 int ycol0, long long M, int C) {
    const long long total = M * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C;
        const int c = (int)(i - m * C);
        Y[m * ldy + ycol0 + c] = X[m * ldx + xcol0 + c];
    }
}

// Y[m, ycol0 : ycol0 + C] = X[idx[m], xcol0 : xcol0 + C]: one thread per 16-byte chunk (C % 8 == 0, 16-byte aligned rows)
__global__ void gather_rows_kernel(const half_t* __restrict__ X, int ldx, int xcol0, const int* __restrict__ idx,
                                   half_t* __restrict__ Y, int ldy, int ycol0, long long M, int C8) {
    const long long total = M * C8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C8;
        const int c = (int)(i - m * C8) * 8;
        *(h8*)(Y + m * ldy + ycol0 + c) = *(const h8*)(X + (long long)idx[m] * ldx + xcol0 + c);
    }
}

// Rotary position embedding, in place: the pair (2 i, 2 i + 1) of columns [col0, col0 + rot_dim) of row r is rotated by the
// angle pos(r) * theta^(-2 i / rot_dim), pos(r) = (r / rows_per_pos) % n_pos.  One thread per 4 pairs (16 bytes).
__global__ void rotary_kernel(half_t* __restrict__ X, int ld, long long rows, int col0, int rot_dim, int n_win, int win_stride,
                              int rows_per_pos, int n_pos, float log2_theta) {
    const int C8 = rot_dim >> 3, W8 = C8 * n_win;
    const long long total = rows * W8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / W8;
        const int wc = (int)(i - r * W8);
        const int win = wc / C8, c = (wc - win * C8) * 8;
        const float pos = (float)((r / rows_per_pos) % n_pos);
        half_t* px = X + r * ld + col0 + win * win_stride + c;
        h8 v = *(const h8*)px;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float freq = exp2f(-log2_theta * (float)(c + 2 * e) / (float)rot_dim);
            float sn, cs;
            sincosf(pos * freq, &sn, &cs);
            const float a = (float)v[2 * e], b = (float)v[2 * e + 1];
            v[2 * e] = (half_t)(a * cs - b * sn);
            v[2 * e + 1] = (half_t)(b * cs + a * sn);
        }
        *(h8*)px = v;
    }
}

__global__ void cfg_ddim_step_kernel(const half_t* __restrict__ V, int ldv, int b_unc, int b_cond, float g,
                                     const float* __restrict__ coef, const half_t* __restrict__ lat,
                                     half_t* __restrict__ out, int C, int F, int HW) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= F * HW) return;
    const float sa_t = coef[0], sb_t = coef[1], sa_p = coef[2], sb_p = coef[3];
    const long long rc = ((long long)b_cond * F * HW + idx) * ldv;
    const long long ru = ((long long)(b_unc < 0 ? 0 : b_unc) * F * HW + idx) * ldv;
    const int f = idx / HW, p = idx - f * HW;
    for (int c = 0; c < C; ++c) {
        float v = (float)V[rc + c];
        if (b_unc >= 0) {
            // fp16 CFG arithmetic order of the reference: neg + g * (edit - neg), each op rounded to fp16
            const float vu = (float)V[ru + c];
            const float diff = (float)(half_t)(v - vu);
            v = (float)(half_t)(vu + (float)(half_t)(g * diff));
        }
        const long long li = ((long long)c * F + f) * HW + p;
        const float x = (float)lat[li];
        const float x0 = sa_t * x - sb_t * v;
        const float eps = sa_t * v + sb_t * x;
        out[li] = (half_t)(sa_p * x0 + sb_p * eps);
    }
}

__global__ void ddim_step_kernel(const half_t* __restrict__ V, const half_t* __restrict__ X, half_t* __restrict__ Y,
                                 float sa_t, float sb_t, float sa_p, float sb_p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = (float)V[i], x = (float)X[i];
        const float x0 = sa_t * x - sb_t * v;
        const float eps = sa_t * v + sb_t * x;
        Y[i] = (half_t)(sa_p * x0 + sb_p * eps);
    }
}

// Guidance + DDIM step on same-layout tensors (the ConsistI2V loop): E holds the UNet's prediction of every branch, n elements each.
//   e = e_unc + g_img * (e_img - e_unc) + g_txt * (e_txt - e_img)      (b_img < 0: e_unc + g_txt * (e_txt - e_unc); b_unc < 0: e_txt)
// each guidance operation rounded to fp16 like the reference's fp16 tensors; the step in fp32.
__global__ void guided_step_kernel(const half_t* __restrict__ E, long long n, int b_unc, int b_img, int b_txt, float g_img, float g_txt,
                                   int pred, float sa_t, float sb_t, float sa_p, float sb_p, const half_t* __restrict__ X,
                                   half_t* __restrict__ Y, const half_t* __restrict__ N, float sigma) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float e = (float)E[(long long)b_txt * n + i];
        if (b_unc >= 0) {
            const float eu = (float)E[(long long)b_unc * n + i];
            if (b_img >= 0) {
                const float ei = (float)E[(long long)b_img * n + i];
                const float a = (float)(half_t)(g_img * (float)(half_t)(ei - eu));
                const float b = (float)(half_t)(g_txt * (float)(half_t)(e - ei));
                e = (float)(half_t)((float)(half_t)(eu + a) + b);
            } else {
                e = (float)(half_t)(eu + (float)(half_t)(g_txt * (float)(half_t)(e - eu)));
            }
        }
        const float x = (float)X[i];
        float x0, eps;
        if (pred == 0) {          // v_prediction
            x0 = sa_t * x - sb_t * e;
            eps = sa_t * e + sb_t * x;
        } else if (pred == 1) {   // epsilon
            x0 = (x - sb_t * e) / sa_t;
            eps = e;
        } else {                  // sample
            x0 = e;
            eps = (x - sa_t * e) / sb_t;
        }
        float y = sa_p * x0 + sb_p * eps;
        if (N != nullptr) y += sigma * (float)N[i];   // ancestral (DDPM) step: + sqrt(variance) * noise
        Y[i] = (half_t)y;
    }
}

static inline unsigned nblk(long long n, int t, long long cap = 65535) {
    long long b = (n + t - 1) / t;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

// Row softmax of fp32 logits -> fp16 probabilities: P[r, :] = softmax(scale * S[r, :]).  One block per row, the row
// lives in registers (cols <= 256 * 32).  Used by the single 512-wide attention head of the AutoencoderKL mid block
// (logits come from anyv2v_gemm_f16 with act = 4).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int lds_, half_t* __restrict__ P,
                                                           int ldp, int cols, float scale_log2) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* src = S + (size_t)row * lds_;
    float v[32];
    float mx = -1e30f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int c = i * 256 + tid;
        v[i] = c < cols ? src[c] * scale_log2 : -1e30f;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        v[i] = i * 256 + tid < cols ? __builtin_amdgcn_exp2f(v[i] - mx) : 0.f;
        sum += v[i];
    }
    sum = wave_sum(sum);
    if (lane == 0) red[w] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    half_t* dst = P + (size_t)row * ldp;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int c = i * 256 + tid;
        if (c < cols) dst[c] = (half_t)(v[i] * inv);
    }
}

extern "C" int anyv2v_silu_f16(const void* X, void* Y, int64_t n, void* stream) {
    AV_CHECK(X && Y && n > 0, "silu: bad arguments");
    hipLaunchKernelGGL(silu_kernel, dim3(nblk(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const half_t*)X,
                       (half_t*)Y, (long long)n);
    return av_launch_status("silu");
}

extern "C" int anyv2v_add_f16(const void* A, const void* B, void* Y, int64_t n, void* stream) {
    AV_CHECK(A && B && Y && n > 0, "add: bad arguments");
    const bool vec = av_aligned16(A) && av_aligned16(B) && av_aligned16(Y);
    const long long n8 = vec ? n / 8 : 0;
    hipLaunchKernelGGL(add_kernel, dim3(nblk(n / 8 + 1, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const half_t*)A,
                       (const half_t*)B, (half_t*)Y, n8, (long long)n);
    return av_launch_status("add");
}

extern "C" int anyv2v_timestep_embedding_f16(const float* t, void* out, int32_t B, int32_t dim, void* stream) {
    AV_CHECK(t && out && B > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: bad arguments");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(nblk((long long)B * dim, 256)), dim3(256), 0, (hipStream_t)stream,
                       t, (half_t*)out, B, dim);
    return av_launch_status("timestep_embedding");
}

extern "C" int anyv2v_ncfhw_to_tokens_f16(const void* X, void* Y, int32_t B, int32_t C, int32_t F, int32_t HW,
                                          int32_t ldy, int32_t col0, void* stream) {
    AV_CHECK(X && Y && B > 0 && C > 0 && F > 0 && HW > 0 && ldy >= col0 + C, "ncfhw_to_tokens: bad arguments");
    hipLaunchKernelGGL(ncfhw_to_tokens_kernel, dim3(nblk((long long)B * F * HW, 256, 1 << 30)), dim3(256), 0,
                       (hipStream_t)stream, (const half_t*)X, (half_t*)Y, B, C, F, HW, ldy, col0);
    return av_launch_status("ncfhw_to_tokens");
}

extern "C" int anyv2v_tokens_to_ncfhw_f16(const void* X, void* Y, int32_t B, int32_t C, int32_t F, int32_t HW,
                                          int32_t ldx, int32_t col0, void* stream) {
    AV_CHECK(X && Y && B > 0 && C > 0 && F > 0 && HW > 0 && ldx >= col0 + C, "tokens_to_ncfhw: bad arguments");
    hipLaunchKernelGGL(tokens_to_ncfhw_kernel, dim3(nblk((long long)B * F * HW, 256, 1 << 30)), dim3(256), 0,
                       (hipStream_t)stream, (const half_t*)X, (half_t*)Y, B, C, F, HW, ldx, col0);
    return av_launch_status("tokens_to_ncfhw");
}

extern "C" int anyv2v_adaptive_avgpool_f16(const void* X, void* Y, int32_t N, int32_t Hi, int32_t Wi, int32_t Ho,
                                           int32_t Wo, int32_t C, void* stream) {
    AV_CHECK(X && Y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "adaptive_avgpool: bad arguments");
    hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3(nblk((long long)N * Ho * Wo * C, 256, 1 << 30)), dim3(256), 0,
                       (hipStream_t)stream, (const half_t*)X, (half_t*)Y, N, Hi, Wi, Ho, Wo, C);
    return av_launch_status("adaptive_avgpool");
}

extern "C" int anyv2v_copy_cols_f16(const void* X, int32_t ldx, int32_t xcol0, void* Y, int32_t ldy, int32_t ycol0,
                                    int64_t M, int32_t C, void* stream) {
    AV_CHECK(X && Y && M > 0 && C > 0, "copy_cols: bad arguments");
    hipLaunchKernelGGL(copy_cols_kernel, dim3(nblk(M * C, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)X, ldx, xcol0, (half_t*)Y, ldy, ycol0, (long long)M, C);
    return av_launch_status("copy_cols");
}

extern "C" int anyv2v_gather_rows_f16(const void* X, int32_t ldx, int32_t xcol0, const int32_t* idx, void* Y, int32_t ldy,
                                      int32_t ycol0, int64_t M, int32_t C, void* stream) {
    AV_CHECK(X && Y && idx && M > 0 && C > 0, "gather_rows: bad arguments");
    AV_CHECK(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && xcol0 % 8 == 0 && ycol0 % 8 == 0 && av_aligned16(X) && av_aligned16(Y),
             "gather_rows: rows and column windows must be 16-byte aligned");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nblk(M * (C / 8), 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)X, ldx, xcol0, (const int*)idx, (half_t*)Y, ldy, ycol0, (long long)M, C / 8);
    return av_launch_status("gather_rows");
}

extern "C" int anyv2v_rotary_f16(void* X, int32_t ld, int64_t rows, int32_t col0, int32_t rot_dim, int32_t n_windows,
                                 int32_t window_stride, int32_t rows_per_pos, int32_t n_pos, float theta, void* stream) {
    AV_CHECK(X && rows > 0 && rot_dim > 0 && n_windows > 0 && rows_per_pos > 0 && n_pos > 0 && theta > 0.f, "rotary: bad arguments");
    AV_CHECK(rot_dim % 8 == 0 && ld % 8 == 0 && col0 % 8 == 0 && window_stride % 8 == 0 && (n_windows == 1 || window_stride >= rot_dim) &&
                 col0 + (n_windows - 1) * window_stride + rot_dim <= ld && av_aligned16(X),
             "rotary: the rotated column windows must be 16-byte aligned, disjoint and inside the row");
    hipLaunchKernelGGL(rotary_kernel, dim3(nblk(rows * (rot_dim / 8) * n_windows, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (half_t*)X, ld, (long long)rows, col0, rot_dim, n_windows, window_stride, rows_per_pos, n_pos, log2f(theta));
    return av_launch_status("rotary");
}

extern "C" int anyv2v_softmax_rows_f32_f16(const float* S, int32_t lds_, void* P, int32_t ldp, int32_t rows, int32_t cols,
                                           float scale, void* stream) {
    AV_CHECK(S && P && rows > 0 && cols > 0 && cols <= 8192 && lds_ >= cols && ldp >= cols, "softmax_rows: bad arguments");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, S, lds_, (half_t*)P, ldp,
                       cols, scale * 1.4426950408889634f);
    return av_launch_status("softmax_rows");
}

extern "C" int anyv2v_cfg_ddim_step_f16(const void* Vtok, int32_t ldv, int32_t b_unc, int32_t b_cond, float guidance,
                                        const float* coef, const void* lat, void* out, int32_t C, int32_t F,
                                        int32_t HW, void* stream) {
    AV_CHECK(Vtok && coef && lat && out && C > 0 && F > 0 && HW > 0 && ldv >= C && b_cond >= 0,
             "cfg_ddim_step: bad arguments");
    hipLaunchKernelGGL(cfg_ddim_step_kernel, dim3(nblk((long long)F * HW, 256, 1 << 30)), dim3(256), 0,
                       (hipStream_t)stream, (const half_t*)Vtok, ldv, b_unc, b_cond, guidance, coef, (const half_t*)lat,
                       (half_t*)out, C, F, HW);
    return av_launch_status("cfg_ddim_step");
}

extern "C" int anyv2v_ddim_step_f16(const void* V, const void* X, void* Y, float sa_t, float sb_t, float sa_p,
                                    float sb_p, int64_t n, void* stream) {
    AV_CHECK(V && X && Y && n > 0, "ddim_step: bad arguments");
    hipLaunchKernelGGL(ddim_step_kernel, dim3(nblk(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const half_t*)V,
                       (const half_t*)X, (half_t*)Y, sa_t, sb_t, sa_p, sb_p, (long long)n);
    return av_launch_status("ddim_step");
}

extern "C" int anyv2v_guided_step_f16(const void* E, int64_t n, int32_t b_unc, int32_t b_img, int32_t b_txt, float g_img, float g_txt,
                                      int32_t prediction, float sa_t, float sb_t, float sa_p, float sb_p, const void* X, void* Y,
                                      void* stream) {
    AV_CHECK(E && X && Y && n > 0 && b_txt >= 0 && prediction >= 0 && prediction <= 2 && (b_img < 0 || b_unc >= 0),
             "guided_step: bad arguments");
    AV_CHECK(!(prediction == 1 && sa_t == 0.f) && !(prediction == 2 && sb_t == 0.f), "guided_step: this prediction type is singular at this alpha");
    hipLaunchKernelGGL(guided_step_kernel, dim3(nblk(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const half_t*)E, (long long)n,
                       b_unc, b_img, b_txt, g_img, g_txt, prediction, sa_t, sb_t, sa_p, sb_p, (const half_t*)X, (half_t*)Y,
                       (const half_t*)nullptr, 0.f);
    return av_launch_status("guided_step");
}

extern "C" int anyv2v_guided_step_noise_f16(const void* E, int64_t n, int32_t b_unc, int32_t b_img, int32_t b_txt, float g_img, float g_txt,
                                            int32_t prediction, float sa_t, float sb_t, float c_x0, float c_eps, const void* X, void* Y,
                                            const void* noise, float sigma, void* stream) {
    AV_CHECK(E && X && Y && n > 0 && b_txt >= 0 && prediction >= 0 && prediction <= 2 && (b_img < 0 || b_unc >= 0) && (noise || sigma == 0.f),
             "guided_step_noise: bad arguments");
    AV_CHECK(!(prediction == 1 && sa_t == 0.f) && !(prediction == 2 && sb_t == 0.f), "guided_step_noise: this prediction type is singular at this alpha");
    hipLaunchKernelGGL(guided_step_kernel, dim3(nblk(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const half_t*)E, (long long)n,
                       b_unc, b_img, b_txt, g_img, g_txt, prediction, sa_t, sb_t, c_x0, c_eps, (const half_t*)X, (half_t*)Y,
                       (const half_t*)noise, sigma);
    return av_launch_status("guided_step_noise");
}

// ---------------------------------------------------------------------------------------------------------
// Layout self-test: runs single MFMAs / an LDS transpose read with the lane<->element maps the kernels assume and
// dumps the results; the gpu test-suite checks them against torch matmuls (asymmetric operands, per the guide).
// scratch layout (bytes): A16[16x32 h] @0 | B16[32x16 h] @1024 | D16[16x16 f32] @2048 | A32[32x16 h] @3072 |
//                         B32[16x32 h] @4096 | D32[32x32 f32] @5120 | TR[64 x 4 h] @9216 | (end 9728)
typedef __fp16 fp16x4_t __attribute__((__vector_size__(8)));

__global__ void selftest_kernel(char* s) {
    __shared__ __attribute__((aligned(16))) half_t lds[1024];
    const int l = threadIdx.x;
    const half_t* A16 = (const half_t*)(s + 0);
    const half_t* B16 = (const half_t*)(s + 1024);
    float* D16 = (float*)(s + 2048);
    const half_t* A32 = (const half_t*)(s + 3072);
    const half_t* B32 = (const half_t*)(s + 4096);
    float* D32 = (float*)(s + 5120);
    half_t* TR = (half_t*)(s + 9216);
    {  // 16x16x32: a lane = A[l&15][8(l>>4)+j], b lane = B[8(l>>4)+j][l&15]; D[4(l>>4)+r][l&15]
        h8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = A16[(l & 15) * 32 + 8 * (l >> 4) + j];
            b[j] = B16[(8 * (l >> 4) + j) * 16 + (l & 15)];
        }
        f4 d = {0.f, 0.f, 0.f, 0.f};
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D16[(4 * (l >> 4) + r) * 16 + (l & 15)] = d[r];
    }
    {  // 32x32x16: a lane = A[l&31][8(l>>5)+j], b lane = B[8(l>>5)+j][l&31]; D[(r&3)+8(r>>2)+4(l>>5)][l&31]
        h8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = A32[(l & 31) * 16 + 8 * (l >> 5) + j];
            b[j] = B32[(8 * (l >> 5) + j) * 32 + (l & 31)];
        }
        f16v d;
        for (int r = 0; r < 16; ++r) d[r] = 0.f;
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
        for (int r = 0; r < 16; ++r) D32[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = d[r];
    }
    {  // ds_read_b64_tr_b16 probe: lds[i] = i, lane l reads from &lds[4 l]
        for (int i = l; i < 1024; i += 64) lds[i] = (half_t)(float)i;
        __syncthreads();
        fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(lds + l * 4));
        for (int j = 0; j < 4; ++j) TR[l * 4 + j] = (half_t)v[j];
    }
}

extern "C" int anyv2v_selftest(void* scratch, int64_t scratch_bytes, void* stream) {
    AV_CHECK(scratch != nullptr && scratch_bytes >= 9728, "selftest: scratch must be >= 9728 bytes");
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (char*)scratch);
    return av_launch_status("selftest");
}
