// K-tile microbenchmark (gfx950): one persistent block per CU, 8 waves; per "K-tile" every wave runs NMFMA MFMAs (16x16x32 f16) in
// groups of 3 with NREAD 16-byte fragment reads (ds_read_b128, conflict-free swizzled rows) and NDMA 1-KiB LDS-DMA pieces threaded
// through the groups -- the instruction mix of gemm_big_kernel's K loop without its address arithmetic -- then synchronises:
//   SYNC 1: s_waitcnt vmcnt(0) + s_barrier            (2-stage ring: everything requested during this K-tile must have landed)
//   SYNC 2: s_waitcnt vmcnt(NDMA) + s_barrier         (3-stage ring: only the previous K-tile's pieces must have landed)
//   SYNC 0: nothing (free running)
// Prints shader-clock-independent ticks (s_memtime) per K-tile and per MFMA per SIMD.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ktile_probe tools/ktile_probe.hip && /tmp/ktile_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void vm_wait() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

template <int NDMA, int NMFMA, int NREAD, int SYNC, int ROWB>
__global__ __launch_bounds__(512) void probe(const _Float16* src, long long* out, int iters, int lds_pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nw = blockDim.x >> 6;
    // LDS: two 64-KiB areas written alternately by the DMA (wave w: KiB w, w + 8, ...); fragments are read from a 48-KiB
    // window that moves over three positions: rows of ROWB bytes (128: 64-deep K-tiles, 64: 32-deep), 16-byte chunk XOR-swizzled by row
    const int l15 = lane & 15, lq = lane >> 4;
    unsigned frag_off;
    if constexpr (ROWB == 128)
        frag_off = l15 * 128 + ((lq ^ (l15 & 7)) * 16);
    else {
        const int pi[4] = {0, 3, 1, 2};
        frag_off = l15 * 64 + ((((l15 >> 2) & 3) ^ pi[lq]) * 16);
    }
    const _Float16* base = src + ((size_t)(blockIdx.x & 63) * 512 * 1024) + (size_t)(w * 64 + lane) * 8;
    constexpr int NG = NMFMA / 3;
    f4 acc[10][3];
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    h8 a[3], b[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[0][e] = (_Float16)(float)(lane + e);
        a[1][e] = (_Float16)(float)(lane - e);
        a[2][e] = (_Float16)(float)(e);
        b[0][e] = (_Float16)(float)(lane * e);
        b[1][e] = (_Float16)(float)(1 + e);
    }
    if (lds_pad < 0) {  // operands from the (random) source buffer instead of small integers: data-dependent MFMA power
#pragma unroll
        for (int j = 0; j < 3; ++j) a[j] = *(const h8*)(src + (size_t)(j * 64 + lane) * 8 + 4096 * w);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *(const h8*)(src + (size_t)((3 + j) * 64 + lane) * 8 + 4096 * w);
        lds_pad = 0;
    }
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    int st_r = 1;
    for (int it = 0; it < iters; ++it) {
        const _Float16* s = base + (size_t)(it & 31) * 8192;
        char* wst = smem + (it & 1) * 65536 + w * 1024;
        const unsigned rbase = (unsigned)(size_t)(smem + st_r * 32768) + frag_off;
        int ndma = 0, nread = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[g % 10][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[g & 1], a[j], acc[g % 10][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // reads spread evenly over the groups; DMA pieces one per group from the first group on
            while (nread * NG < (g + 1) * NREAD) {
                h8 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(rbase), "n"((nread % 23) * 2048) : "memory");
                asm volatile("" ::"v"(v));
                ++nread;
            }
            if (ndma < NDMA) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + ndma * 4096),
                                                 (__attribute__((address_space(3))) void*)(wst + ndma * 8192), 16, 0, 0);
                ++ndma;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (SYNC == 1) {
            vm_wait<0>();
            __builtin_amdgcn_s_barrier();
        } else if constexpr (SYNC == 2) {
            vm_wait<NDMA>();
            __builtin_amdgcn_s_barrier();
        }
        st_r = st_r == 2 ? 0 : st_r + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 12345.678f) out[0] = (long long)smem[lane + lds_pad];
    if (lane == 0) out[1 + blockIdx.x * nw + w] = t1 - t0;
}

template <int NDMA, int NMFMA, int NREAD, int SYNC, int ROWB = 128>
static void run(const char* what, const _Float16* src, long long* out, long long* host, int rnd = 0) {
    const int iters = 400, blocks = 256, waves = 8;
    const size_t lds = 2 * 65536;
    CK(hipFuncSetAttribute((const void*)probe<NDMA, NMFMA, NREAD, SYNC, ROWB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe<NDMA, NMFMA, NREAD, SYNC, ROWB>), dim3(blocks), dim3(waves * 64), lds, 0, src, out, iters, rnd ? -1 : 0);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(host, out, sizeof(long long) * (1 + blocks * waves), hipMemcpyDeviceToHost));
    double s = 0;
    for (int i = 0; i < blocks * waves; ++i) s += (double)host[1 + i];
    const double ticks = s / (blocks * waves) / iters;
    const double tf = 2.0 * 16 * 16 * 32 * NMFMA * 8.0 * blocks * iters / (ms * 1e-3) / 1e12;
    printf("%-34s dma=%d mfma=%2d reads=%2d rows=%3dB sync=%d : %7.0f ticks/K-tile  %5.1f ticks/MFMA/SIMD  %7.1f ns/K-tile  %6.0f TF/s\n", what,
           NDMA, NMFMA, NREAD, ROWB, SYNC, ticks, ticks / (2.0 * NMFMA), ms * 1e6 / iters, tf);
    fflush(stdout);
}


// ---- operand-data sweep: back-to-back MFMAs only (no LDS, no DMA, no barrier), 8 waves per CU on all 256 CUs -------------------
// OPS: 0 = all-zero operands, 1 = small integers, 2 = random fp16 in [-2, 2).  SHAPE: 0 = 16x16x32, 1 = 32x32x16.
typedef float f16v __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_only(const _Float16* src, long long* out, int iters, int ops) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    h8 a[3], b[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[0][e] = ops == 1 ? (_Float16)(float)(lane + e) : (_Float16)0.f;
        a[1][e] = ops == 1 ? (_Float16)(float)(lane - e) : (_Float16)0.f;
        a[2][e] = ops == 1 ? (_Float16)(float)(e) : (_Float16)0.f;
        b[0][e] = ops == 1 ? (_Float16)(float)(lane * e) : (_Float16)0.f;
        b[1][e] = ops == 1 ? (_Float16)(float)(1 + e) : (_Float16)0.f;
    }
    if (ops == 2) {
#pragma unroll
        for (int j = 0; j < 3; ++j) a[j] = *(const h8*)(src + (size_t)(j * 64 + lane) * 8 + 4096 * w);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *(const h8*)(src + (size_t)((3 + j) * 64 + lane) * 8 + 4096 * w);
    }
    float sum = 0.f;
    if constexpr (SHAPE == 0) {
        f4 acc[10][3];
#pragma unroll
        for (int i = 0; i < 10; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 20; ++g)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[g % 10][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[g & 1], a[j], acc[g % 10][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 10; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) sum += acc[i][j][0] + acc[i][j][3];
    } else {
        f16v acc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 30; ++g) acc[g % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[g & 1], a[g % 3], acc[g % 6], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) sum += acc[i][0] + acc[i][15];
    }
    if (sum == 12345.678f) out[0] = 1;
}

template <int SHAPE>
static void run_ops(const _Float16* src, long long* out, int ops) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((mfma_only<SHAPE>), dim3(blocks), dim3(512), 0, 0, src, out, iters, ops);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    // per SIMD: 2 waves x 60 MFMAs (16 cycles each) or 2 x 30 (32 cycles each) per iteration = 1920 matrix-pipe cycles
    const double flops = (SHAPE == 0 ? 2.0 * 16 * 16 * 32 * 60 : 2.0 * 32 * 32 * 16 * 30) * 8.0 * blocks * iters;
    const double ghz = 1920.0 * iters / (ms * 1e-3) / 1e9;
    printf("MFMA only %-9s operands %-14s : %7.1f TF/s  (matrix pipe busy 100 %% => clock %.2f GHz)\n", SHAPE == 0 ? "16x16x32" : "32x32x16",
           ops == 0 ? "all zero" : (ops == 1 ? "small integers" : "random fp16"), flops / (ms * 1e-3) / 1e12, ghz);
    fflush(stdout);
}

// ---- one wave per SIMD (4 waves, 512 registers each: accumulators can live in AGPRs), 32x32x16 MFMAs: the 192 x 320 tile as a
//      2 x 2 grid of 96 x 160 wave tiles = 15 accumulator tiles; per K-tile and wave 60 MFMAs (32 cycles each), 32 fragment reads,
//      16 LDS-DMA pieces.  SYNC as above.
template <int NDMA, int NREAD, int SYNC>
__global__ __launch_bounds__(256) void probe4(const _Float16* src, long long* out, int iters, int rnd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned frag_off = l15 * 128 + ((lq ^ (l15 & 7)) * 16);
    const _Float16* base = src + ((size_t)(blockIdx.x & 63) * 512 * 1024) + (size_t)(w * 64 + lane) * 8;
    f16v acc[15];
#pragma unroll
    for (int i = 0; i < 15; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a[3], b[5];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[j][e] = (_Float16)(float)(lane + e + j);
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) b[j][e] = (_Float16)(float)(lane - e * j);
    if (rnd) {
#pragma unroll
        for (int j = 0; j < 3; ++j) a[j] = *(const h8*)(src + (size_t)(j * 64 + lane) * 8 + 4096 * w);
#pragma unroll
        for (int j = 0; j < 5; ++j) b[j] = *(const h8*)(src + (size_t)((3 + j) * 64 + lane) * 8 + 4096 * w);
    }
    __syncthreads();
    int st_r = 1;
    for (int it = 0; it < iters; ++it) {
        const _Float16* s = base + (size_t)(it & 31) * 8192;
        char* wst = smem + (it & 1) * 65536 + w * 1024;
        const unsigned rbase = (unsigned)(size_t)(smem + st_r * 32768) + frag_off;
        int ndma = 0, nread = 0;
#pragma unroll
        for (int g = 0; g < 20; ++g) {  // 4 k-steps x 5 column tiles: 3 MFMAs (the row tiles) per group
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[(g % 5) * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[g % 5], a[j], acc[(g % 5) * 3 + j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            while (nread * 20 < (g + 1) * NREAD) {
                h8 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(rbase), "n"((nread % 23) * 2048) : "memory");
                asm volatile("" ::"v"(v));
                ++nread;
            }
            if (ndma < NDMA) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + ndma * 2048),
                                                 (__attribute__((address_space(3))) void*)(wst + ndma * 4096), 16, 0, 0);
                ++ndma;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (SYNC == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else if constexpr (SYNC == 2) {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        st_r = st_r == 2 ? 0 : st_r + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 15; ++i) sum += acc[i][0] + acc[i][15];
    if (sum == 12345.678f) out[0] = (long long)smem[lane];
}

template <int NDMA, int NREAD, int SYNC>
static void run4(const char* what, const _Float16* src, long long* out, int rnd) {
    const int iters = 400, blocks = 256;
    const size_t lds = 2 * 65536;
    CK(hipFuncSetAttribute((const void*)probe4<NDMA, NREAD, SYNC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe4<NDMA, NREAD, SYNC>), dim3(blocks), dim3(256), lds, 0, src, out, iters, rnd);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double tf = 2.0 * 32 * 32 * 16 * 60 * 4.0 * blocks * iters / (ms * 1e-3) / 1e12;
    printf("%-44s dma=%2d mfma=60 (32x32x16) reads=%2d sync=%d %s : %7.1f ns/K-tile  %6.0f TF/s\n", what, NDMA, NREAD, SYNC,
           rnd ? "random ops" : "small ints", ms * 1e6 / iters, tf);
    fflush(stdout);
}

int main() {
    _Float16* src;
    long long* out;
    CK(hipMalloc(&src, (size_t)64 * 1024 * 1024 + (1 << 20)));
    {   // random fp16 in [-2, 2) (uniform bit patterns in the mantissa: realistic toggling for the DMA and the MFMA operands)
        const size_t n = ((size_t)64 * 1024 * 1024 + (1 << 20)) / 2;
        _Float16* h = (_Float16*)malloc(n * 2);
        unsigned x = 12345u;
        for (size_t i = 0; i < n; ++i) {
            x = x * 1664525u + 1013904223u;
            h[i] = (_Float16)(((int)(x >> 8 & 0xffff) - 32768) / 16384.0f);
        }
        CK(hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice));
        free(h);
    }
    CK(hipMalloc(&out, sizeof(long long) * (1 + 256 * 8)));
    long long* host = (long long*)malloc(sizeof(long long) * (1 + 256 * 8));
    for (int rep = 0; rep < 2; ++rep)
        for (int ops = 0; ops < 3; ++ops) {
            run_ops<0>(src, out, ops);
            run_ops<1>(src, out, ops);
        }
    for (int rnd = 0; rnd < 2; ++rnd) {
        run4<0, 0, 0>("1 wave/SIMD, MFMA only, free", src, out, rnd);
        run4<0, 32, 0>("1 wave/SIMD, MFMA + reads, free", src, out, rnd);
        run4<16, 32, 0>("1 wave/SIMD, 192x320 mix, free", src, out, rnd);
        run4<16, 32, 1>("1 wave/SIMD, 192x320 mix, drain", src, out, rnd);
        run4<16, 32, 2>("1 wave/SIMD, 192x320 mix, ring", src, out, rnd);
        run<8, 60, 26, 1>("2 waves/SIMD 192x320 mix, drain (today)", src, out, host, rnd);
    }
    run<0, 60, 0, 0>("MFMA only, free", src, out, host);
    run<0, 60, 0, 0>("MFMA only, free, random operands", src, out, host, 1);
    run<8, 60, 26, 1>("192x320 mix, drain, random ops", src, out, host, 1);
    run<8, 60, 26, 2>("192x320 mix, ring, random ops", src, out, host, 1);
    run<7, 30, 16, 2>("96x320 mix, ring, random ops", src, out, host, 1);
    run<0, 60, 0, 2>("MFMA only, barrier", src, out, host);
    run<0, 60, 26, 0>("MFMA + reads, free", src, out, host);
    run<0, 60, 26, 2>("MFMA + reads, barrier", src, out, host);
    run<8, 60, 26, 0>("192x320 mix, free", src, out, host);
    run<8, 60, 26, 1>("192x320 mix, drain (today)", src, out, host);
    run<8, 60, 26, 2>("192x320 mix, ring (LDS: no)", src, out, host);
    run<8, 60, 0, 1>("192x320 no reads, drain", src, out, host);
    run<8, 60, 0, 2>("192x320 no reads, ring", src, out, host);
    run<7, 30, 16, 1>("96x320 mix, drain", src, out, host);
    run<7, 30, 16, 2>("96x320 mix, ring", src, out, host);
    run<7, 30, 16, 0>("96x320 mix, free", src, out, host);
    run<7, 39, 18, 2>("256x160 mix, ring (40 MFMA)", src, out, host);
    run<4, 30, 13, 1, 64>("192x320 32-deep, drain", src, out, host);
    run<4, 30, 13, 2, 64>("192x320 32-deep, 5-slot ring", src, out, host);
    run<8, 60, 26, 2, 64>("192x320 2x32-deep per barrier", src, out, host);
    return 0;
}
