// Microbenchmark (gfx950): do MFMA and VALU instructions of one SIMD overlap, and what do the softmax's VALU instructions
// cost?  Every wave runs ITERS iterations of a body built from inline-asm instructions (nothing for the compiler to
// reorder or remove); 1, 2 or 3 waves per SIMD.  Reported: ns per body per SIMD (kernel time / iterations / waves per
// SIMD) -- if MFMA and VALU overlapped, body "M+V" would cost max(M, V), not M + V.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_probe tools/issue_probe.hip && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))
#define ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c2))
#define MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(p1), "v"(p2))
#define PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(p2))
#define PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(p1))
#define CVT(d, x, y) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define DOT2(x, hh) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x) : "v"(hh), "v"(hone))

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* in, float* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)in[(tid + i) & 1023];
        b[i] = (_Float16)in[(tid * 3 + i) & 1023];
    }
    f16v acc[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    float x[8];
    f2v px[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = in[(tid + 64 * i) & 1023];
        px[i] = f2v{x[i], x[i] * 0.5f};
    }
    const float c1 = 0.999f, c2 = -0.001f;
    const f2v p1 = {0.999f, 0.998f}, p2 = {-0.001f, -0.002f};
    unsigned hcv[4] = {0, 0, 0, 0};
    const unsigned hone = 0x3c003c00u, hval = 0x2c002c00u;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 16 MFMA
#pragma unroll
            for (int g = 0; g < 16; ++g) MFMA(acc[g & 3]);
        } else if (MODE == 1) {  // 128 v_fma
#pragma unroll
            for (int g = 0; g < 128; ++g) FMA(x[g & 7]);
        } else if (MODE == 2) {  // 64 v_pk_fma
#pragma unroll
            for (int g = 0; g < 64; ++g) PKFMA(px[g & 7]);
        } else if (MODE == 3) {  // 32 v_exp
#pragma unroll
            for (int g = 0; g < 32; ++g) EXP(x[g & 7]);
        } else if (MODE == 4) {  // 16 MFMA then 128 v_fma
#pragma unroll
            for (int g = 0; g < 16; ++g) MFMA(acc[g & 3]);
#pragma unroll
            for (int g = 0; g < 128; ++g) FMA(x[g & 7]);
        } else if (MODE == 5) {  // (1 MFMA + 8 v_fma) x 16
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                MFMA(acc[g & 3]);
#pragma unroll
                for (int e = 0; e < 8; ++e) FMA(x[e]);
            }
        } else if (MODE == 6) {  // (1 MFMA + 4 v_pk_fma) x 16
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                MFMA(acc[g & 3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) PKFMA(px[e + 4 * (g & 1)]);
            }
        } else if (MODE == 7) {  // (1 MFMA + 2 v_exp + 6 v_fma) x 16
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                MFMA(acc[g & 3]);
                EXP(x[0 + 2 * (g & 3)]);
                EXP(x[1 + 2 * (g & 3)]);
#pragma unroll
                for (int e = 0; e < 6; ++e) FMA(x[(e + 2 + 2 * (g & 3)) & 7]);
            }
        } else if (MODE == 8) {  // 16 v_max3
#pragma unroll
            for (int g = 0; g < 16; ++g) MAX3(x[g & 7]);
        } else if (MODE == 9) {  // 32 v_add
#pragma unroll
            for (int g = 0; g < 32; ++g) ADD(x[g & 7]);
        } else if (MODE == 10) {  // 16 v_pk_add
#pragma unroll
            for (int g = 0; g < 16; ++g) PKADD(px[g & 7]);
        } else if (MODE == 11) {  // 16 v_cvt_pk_f16_f32
#pragma unroll
            for (int g = 0; g < 16; ++g) CVT(hcv[g & 3], x[g & 7], x[(g + 1) & 7]);
        } else if (MODE == 12) {  // 16 v_dot2_f32_f16
#pragma unroll
            for (int g = 0; g < 16; ++g) DOT2(x[g & 7], hval);
        } else if (MODE == 13) {  // 16 v_pk_mul
#pragma unroll
            for (int g = 0; g < 16; ++g) PKMUL(px[g & 7]);
        } else if (MODE == 14) {  // softmax-like VALU block alone: 16 max3, 16 pk_fma, 32 exp, 16 pk_add, 16 cvt
#pragma unroll
            for (int g = 0; g < 16; ++g) MAX3(x[g & 7]);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                PKFMA(px[g & 7]);
                EXP(x[(2 * g) & 7]);
                EXP(x[(2 * g + 1) & 7]);
                PKADD(px[(g + 4) & 7]);
                CVT(hcv[g & 3], x[g & 7], x[(g + 1) & 7]);
            }
        } else if (MODE == 15) {  // 16 MFMA then the softmax-like block
#pragma unroll
            for (int g = 0; g < 16; ++g) MFMA(acc[g & 3]);
#pragma unroll
            for (int g = 0; g < 16; ++g) MAX3(x[g & 7]);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                PKFMA(px[g & 7]);
                EXP(x[(2 * g) & 7]);
                EXP(x[(2 * g + 1) & 7]);
                PKADD(px[(g + 4) & 7]);
                CVT(hcv[g & 3], x[g & 7], x[(g + 1) & 7]);
            }
        } else if (MODE == 16) {  // the same, one MFMA in front of each sixth of the block
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                MFMA(acc[g & 3]);
                MAX3(x[g & 7]);
                PKFMA(px[g & 7]);
                EXP(x[(2 * g) & 7]);
                EXP(x[(2 * g + 1) & 7]);
                PKADD(px[(g + 4) & 7]);
                CVT(hcv[g & 3], x[g & 7], x[(g + 1) & 7]);
            }
        }
#define PLAIN10(g)                                                            \
    MAX3(x[(g) & 7]);                                                         \
    FMA(x[((g) + 1) & 7]);                                                    \
    FMA(x[((g) + 2) & 7]);                                                    \
    EXP(x[((g) + 3) & 7]);                                                    \
    EXP(x[((g) + 4) & 7]);                                                    \
    ADD(x[((g) + 5) & 7]);                                                    \
    ADD(x[((g) + 6) & 7]);                                                    \
    CVT(hcv[(g) & 3], x[(g) & 7], x[((g) + 7) & 7]);                          \
    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[((g) + 1) & 7]) : "v"(c1));  \
    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[((g) + 2) & 7]) : "v"(c1))
        else if (MODE == 17) {  // plain-op softmax (no packed math): per MFMA 1 max3, 2 fma, 2 exp, 2 add, 1 cvt_pk, 2 mul
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                MFMA(acc[g & 3]);
                PLAIN10(g);
            }
        } else if (MODE == 18) {  // the same VALU ops alone
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                PLAIN10(g);
            }
        } else if (MODE == 19) {  // 16 MFMA, then the plain-op softmax as one block (the v2 order)
#pragma unroll
            for (int g = 0; g < 16; ++g) MFMA(acc[g & 3]);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                PLAIN10(g);
            }
        }
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    for (int i = 0; i < 8; ++i) s += x[i] + px[i][0] + px[i][1];
    out[tid] = s + (float)(hcv[0] ^ hcv[1] ^ hcv[2] ^ hcv[3]);
}


// ---- per-instruction overlap table: 64 x OP alone, and (1 MFMA + 4 x OP) x 16
#define MAX2(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(c2))
#define MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c1))
#define CVT1(d, x) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(d) : "v"(x))
#define MOV(d, x) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(x))
#define PERM(d, x) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(c2), "v"(c1))
template <int OP>
__device__ __forceinline__ void one_op(float (&x)[8], f2v (&px)[8], unsigned (&hcv)[4], int g, float c1, float c2,
                                       f2v p1, f2v p2, unsigned hone, unsigned hval) {
    if (OP == 0) FMA(x[g & 7]);
    if (OP == 1) ADD(x[g & 7]);
    if (OP == 2) MUL(x[g & 7]);
    if (OP == 3) MAX2(x[g & 7]);
    if (OP == 4) MAX3(x[g & 7]);
    if (OP == 5) EXP(x[g & 7]);
    if (OP == 6) CVT(hcv[g & 3], x[g & 7], x[(g + 1) & 7]);
    if (OP == 7) CVT1(hcv[g & 3], x[g & 7]);
    if (OP == 8) PKFMA(px[g & 7]);
    if (OP == 9) PKADD(px[g & 7]);
    if (OP == 10) PKMUL(px[g & 7]);
    if (OP == 11) DOT2(x[g & 7], hval);
    if (OP == 12) MOV(hcv[g & 3], x[g & 7]);
    if (OP == 13) PERM(hcv[g & 3], x[g & 7]);
}
template <int OP, int WITH_MFMA>
__global__ __launch_bounds__(256) void probe_op(const float* in, float* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)in[(tid + i) & 1023];
        b[i] = (_Float16)in[(tid * 3 + i) & 1023];
    }
    f16v acc[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    float x[8];
    f2v px[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = in[(tid + 64 * i) & 1023];
        px[i] = f2v{x[i], x[i] * 0.5f};
    }
    const float c1 = 0.999f, c2 = -0.001f;
    const f2v p1 = {0.999f, 0.998f}, p2 = {-0.001f, -0.002f};
    unsigned hcv[4] = {0, 0, 0, 0};
    const unsigned hone = 0x3c003c00u, hval = 0x2c002c00u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (WITH_MFMA) MFMA(acc[g & 3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) one_op<OP>(x, px, hcv, 4 * g + e, c1, c2, p1, p2, hone, hval);
        }
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    for (int i = 0; i < 8; ++i) s += x[i] + px[i][0] + px[i][1];
    out[tid] = s + (float)(hcv[0] ^ hcv[1] ^ hcv[2] ^ hcv[3]);
}
static const char* OPN[] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_max_f32", "v_max3_f32", "v_exp_f32",
                            "v_cvt_pk_f16_f32", "v_cvt_f16_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32",
                            "v_dot2_f32_f16", "v_mov_b32", "v_perm_b32"};
template <int OP>
static void run_op(const float* in, float* out) {
    const int iters = 4000, wps = 2;
    float t[2];
    for (int m = 0; m < 2; ++m) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        if (m == 0) hipLaunchKernelGGL((probe_op<OP, 0>), dim3(256 * wps), dim3(256), 0, 0, in, out, 100);
        else hipLaunchKernelGGL((probe_op<OP, 1>), dim3(256 * wps), dim3(256), 0, 0, in, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        if (m == 0) hipLaunchKernelGGL((probe_op<OP, 0>), dim3(256 * wps), dim3(256), 0, 0, in, out, iters);
        else hipLaunchKernelGGL((probe_op<OP, 1>), dim3(256 * wps), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&t[m], e0, e1);
        t[m] = t[m] * 1e6f / iters / wps;
    }
    printf("%-18s 64 alone %7.1f ns per SIMD (%5.2f ns each); (mfma + 4 op) x16 %7.1f ns per SIMD\n", OPN[OP], t[0],
           t[0] / 64, t[1]);
}

// ---- MFMA accumulate chains: 16 MFMAs over NACC accumulators round-robin (1 = every MFMA waits for its predecessor)
template <int NACC>
__global__ __launch_bounds__(256) void probe_chain(const float* in, float* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)in[(tid + i) & 1023];
        b[i] = (_Float16)in[(tid * 3 + i) & 1023];
    }
    f16v acc[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) MFMA(acc[g % NACC]);
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    out[tid] = s;
}
template <int NACC>
static void run_chain(const float* in, float* out) {
    const int iters = 4000;
    for (int wps = 1; wps <= 2; ++wps) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(probe_chain<NACC>, dim3(256 * wps), dim3(256), 0, 0, in, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe_chain<NACC>, dim3(256 * wps), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("16 mfma over %d accumulator(s)      waves/SIMD=%d: %8.1f ns per body per SIMD\n", NACC, wps,
               ms * 1e6 / iters / wps);
    }
}

static const char* NAMES[] = {"16 mfma_32x32x16", "128 v_fma", "64 v_pk_fma", "32 v_exp", "16 mfma ; 128 v_fma",
                              "(mfma + 8 v_fma) x16", "(mfma + 4 v_pk_fma) x16", "(mfma + 2 exp + 6 fma) x16",
                              "16 v_max3", "32 v_add", "16 v_pk_add", "16 v_cvt_pk_f16_f32", "16 v_dot2_f32_f16",
                              "16 v_pk_mul", "softmax block (96 VALU)", "16 mfma ; softmax block",
                              "(mfma + 1/16 softmax block) x16", "(mfma + 10 plain softmax ops) x16",
                              "160 plain softmax ops alone", "16 mfma ; 160 plain softmax ops"};

template <int MODE>
static void run(const float* in, float* out) {
    const int iters = 4000;
    for (int wps = 1; wps <= 3; ++wps) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<MODE>, dim3(256 * wps), dim3(256), 0, 0, in, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<MODE>, dim3(256 * wps), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s waves/SIMD=%d: %8.1f ns per body per wave, %8.1f ns per body per SIMD\n", NAMES[MODE], wps,
               ms * 1e6 / iters, ms * 1e6 / iters / wps);
    }
}

int main(int argc, char** argv) {
    const bool only_chain = argc > 1;
    float *in, *out;
    hipMalloc(&in, 4096);
    hipMalloc(&out, sizeof(float) * 256 * 3 * 256);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = -(float)(i % 37) * 0.037f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    if (argc > 1 && argv[1][0] == 'p') { run<0>(in, out); run<17>(in, out); run<18>(in, out); run<19>(in, out); return 0; }
    if (only_chain) { run_chain<1>(in, out); run_chain<2>(in, out); run_chain<4>(in, out); return 0; }
    run<0>(in, out);  run<1>(in, out);  run<2>(in, out);  run<3>(in, out);  run<4>(in, out);  run<5>(in, out);
    run<6>(in, out);  run<7>(in, out);  run<8>(in, out);  run<9>(in, out);  run<10>(in, out); run<11>(in, out);
    run<12>(in, out); run<13>(in, out); run<14>(in, out); run<15>(in, out); run<16>(in, out);
    run_chain<1>(in, out); run_chain<2>(in, out); run_chain<4>(in, out);
    printf("---- per-op table, 2 waves per SIMD; 16 mfma alone = see first rows\n");
    run_op<0>(in, out);  run_op<1>(in, out);  run_op<2>(in, out);  run_op<3>(in, out);  run_op<4>(in, out);
    run_op<5>(in, out);  run_op<6>(in, out);  run_op<7>(in, out);  run_op<8>(in, out);  run_op<9>(in, out);
    run_op<10>(in, out); run_op<11>(in, out); run_op<12>(in, out); run_op<13>(in, out);
    return 0;
}
