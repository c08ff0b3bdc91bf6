// Shared device/host helpers for the gfx950 kernels (wave64, MFMA f16, fp32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/anyv2v_hip.h"

typedef _Float16 half_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

#define WAVE 64

void anyv2v_set_error(const char* fmt, ...);
int av_hint_rows(int rows);   // rows as the launch heuristics should see them (anyv2v_set_batch_hint), errors.hip

#define AV_CHECK(cond, ...)                 \
    do {                                    \
        if (!(cond)) {                      \
            anyv2v_set_error(__VA_ARGS__);  \
            return ANYV2V_EINVAL;           \
        }                                   \
    } while (0)

static inline int av_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        anyv2v_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return ANYV2V_OK;
}

static inline bool av_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

__device__ __forceinline__ float av_silu(float x) { return x / (1.0f + __expf(-x)); }
// erf-GELU (torch F.gelu default): gelu(x) = x Phi(x) = max(x, 0) - |x| q(|x|) with q = erfc(|x| / sqrt 2) / 2, and erfc by
// Abramowitz-Stegun 7.1.26 (q = 0.5 P(t) t exp(-x^2 / 2), t = 1 / (1 + p |x| / sqrt 2); |error of q| <= 0.75e-7) on the raw
// v_rcp / v_exp instructions.  13 VALU ops -- the GEGLU epilogues evaluate it for every element of the [tokens, 4 dim]
// feed-forward activations and are VALU-issue bound (profiles/r03_gemm_ws_pmc.txt).  The form has no 1 + erf cancellation: over
// all 63 488 finite fp16 inputs 0.41 % of the fp16-rounded results differ from the correctly rounded exact value, by 1 ulp
// (the former 0.5 x (1 + erf) arrangement of the same series: 0.72 %, up to 2 ulp; 16 ops) -- swept on the GPU by
// tests/gpu_checks.py::check_gelu_all_inputs and, with the constants parsed from this file, on the CPU by tests/test_oracle.py.
__device__ __forceinline__ float av_gelu(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, ax, 1.0f));   // p / sqrt 2
    float p = fmaf(0.5307027145f, t, -0.7265760135f);                      // the series' coefficients, halved
    p = fmaf(p, t, 0.7107068705f);
    p = fmaf(p, t, -0.142248368f);
    p = fmaf(p, t, 0.127414796f);
    const float u = x * 0.84932180f;                                       // sqrt(log2(e) / 2): exp2(-u^2) = exp(-x^2 / 2)
    const float q = (p * t) * __builtin_amdgcn_exp2f(-(u * u));
    float m;   // max(x, 0) as ONE instruction: fmaxf() on an MFMA result is preceded by a canonicalising v_max_f32 x, x
    asm("v_max_f32 %0, 0, %1" : "=v"(m) : "v"(x));
    return fmaf(-ax, q, m);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
