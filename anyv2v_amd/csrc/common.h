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
// erf-GELU (torch F.gelu default).  erf via Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, two orders below fp16
// resolution) on the raw v_rcp / v_exp instructions: ~14 VALU ops instead of libm erff's ~35 with branches -- the GEGLU
// epilogue evaluates it for every element of the [tokens, 4*dim] feed-forward activations.
__device__ __forceinline__ float av_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = fmaf(-p * t, e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float av_gelu(float x) { return 0.5f * x * (1.0f + av_erf(x * 0.70710678118654752f)); }

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
