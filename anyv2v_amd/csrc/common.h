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
__device__ __forceinline__ float av_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

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
