// Microbenchmark (gfx950): what does one 1-KiB operand piece cost next to an MFMA stream?
//   mode 0: LDS-DMA (global_load_lds_dwordx4), mode 1: register-staged (global_load_dwordx4 + ds_write_b128)
// Every wave loops: issue n_dma pieces (L2-resident source) -> n_mfma back-to-back MFMAs -> [wait + barrier].
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_probe tools/dma_probe.hip && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int NDMA, int NMFMA, int SYNC>
__global__ __launch_bounds__(512) void probe(const _Float16* src, long long* out, int iters, int lds_pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nw = blockDim.x >> 6;
    char* my = smem + w * (16 * 1024);  // 16 KiB per wave, NDMA <= 16 pieces
    // per-lane source: rows of 128 B, 8 lanes per row (like a K-major tile), block-specific 1 MiB window
    const _Float16* base = src + ((size_t)(blockIdx.x & 63) * 512 * 1024) + (size_t)(w * 64 + lane) * 8;
    f4 acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    h8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)(float)(lane + e);
        b[e] = (_Float16)(float)(lane - e);
    }
    h8 stage[MODE == 1 ? (NDMA > 0 ? NDMA : 1) : 1];
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const _Float16* s = base + (size_t)(it & 31) * 8192;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            if constexpr (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + i * 4096),
                                                 (__attribute__((address_space(3))) void*)(my + i * 1024), 16, 0, 0);
            } else {
                stage[i] = *(const h8*)(s + i * 4096);
            }
        }
#pragma unroll
        for (int i = 0; i < NMFMA; ++i) acc[i % 10] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i % 10], 0, 0, 0);
        if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NDMA; ++i) *(h8*)(my + i * 1024 + lane * 16) = stage[i];
        }
        if constexpr (SYNC == 1) {  // 2-stage ring: drain everything, then barrier
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else if constexpr (SYNC == 2) {  // 3-stage ring: only the previous iteration's pieces must have landed
            if constexpr (MODE == 0) {
                if constexpr (NDMA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                if constexpr (NDMA == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                if constexpr (NDMA == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        } else if constexpr (SYNC == 3) {  // barrier only
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 10; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sum == 12345.678f) out[0] = (long long)smem[lane + lds_pad];  // keep everything alive
    if (lane == 0) out[1 + blockIdx.x * nw + w] = t1 - t0;
}

template <int MODE, int NDMA, int NMFMA, int SYNC>
static void run(const _Float16* src, long long* out, long long* host, int waves) {
    const int iters = 200, blocks = 256;
    const size_t lds = 160 * 1024 - 1024;  // one block per CU
    hipFuncSetAttribute((const void*)probe<MODE, NDMA, NMFMA, SYNC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<MODE, NDMA, NMFMA, SYNC>), dim3(blocks), dim3(waves * 64), lds, 0, src, out, iters, 0);
        hipDeviceSynchronize();
    }
    hipMemcpy(host, out, sizeof(long long) * (1 + blocks * waves), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks * waves; ++i) s += (double)host[1 + i];
    const double cyc = s / (blocks * waves) / iters;
    printf("%-10s waves/SIMD=%d n_dma=%2d n_mfma=%2d sync=%d : %8.0f clk/iter  (MFMA floor %5d clk/wave, x%d waves/SIMD)\n",
           MODE == 0 ? "lds-dma" : "reg-stage", waves / 4, NDMA, NMFMA, (int)SYNC, cyc, NMFMA * 16, waves / 4);
}

#define SWEEP(MODE, SYNC)                                  \
    run<MODE, 0, 40, SYNC>(src, out, host, waves);         \
    run<MODE, 4, 40, SYNC>(src, out, host, waves);         \
    run<MODE, 9, 40, SYNC>(src, out, host, waves);         \
    run<MODE, 16, 40, SYNC>(src, out, host, waves);        \
    run<MODE, 9, 80, SYNC>(src, out, host, waves);         \
    run<MODE, 9, 0, SYNC>(src, out, host, waves);          \
    run<MODE, 16, 0, SYNC>(src, out, host, waves);

int main() {
    _Float16* src;
    long long* out;
    hipMalloc(&src, (size_t)64 * 1024 * 1024 + (1 << 20));
    hipMemset(src, 0, (size_t)64 * 1024 * 1024 + (1 << 20));
    hipMalloc(&out, sizeof(long long) * (1 + 256 * 8));
    long long* host = (long long*)malloc(sizeof(long long) * (1 + 256 * 8));
    for (int waves = 4; waves <= 8; waves += 4) {
        SWEEP(0, 1)
        SWEEP(0, 2)
        SWEEP(0, 3)
        SWEEP(0, 0)
    }
    return 0;
}
