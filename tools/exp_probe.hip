// Microbenchmark (gfx950): can polynomial exp2 on the FMA pipe run beside v_exp_f32 (transcendental, quarter rate)?
//   variant 0: 32 x v_exp_f32 per iteration; variant 1: 32 x polynomial exp2 (floor / fract, degree-4 Horner, ldexp);
//   variant 2: 16 + 16 interleaved.   hipcc --offload-arch=gfx950 -O3 -o /tmp/exp_probe tools/exp_probe.hip && /tmp/exp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ float exp2_poly(float x) {  // x <= 0
    const float xi = floorf(x);
    const float f = x - xi;  // [0, 1)
    float p = 0.0096181291f;
    p = fmaf(p, f, 0.0555041087f);
    p = fmaf(p, f, 0.2402265070f);
    p = fmaf(p, f, 0.6931471806f);
    p = fmaf(p, f, 1.0f);
    return ldexpf(p, (int)xi);
}

template <int VARIANT>
__global__ void probe(const float* in, float* out, long long* ticks, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = in[(tid + i * 64) & 1023];
    float acc = 0.f;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            float v;
            if (VARIANT == 0 || (VARIANT == 2 && (i & 1)))
                v = __builtin_amdgcn_exp2f(x[i]);
            else
                v = exp2_poly(x[i]);
            acc += v;
            x[i] = x[i] * 0.999f - 0.001f;
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    out[tid] = acc;
    if ((threadIdx.x & 63) == 0) ticks[tid >> 6] = t1 - t0;
}

template <int VARIANT>
static void run(const float* in, float* out, long long* ticks, long long* host, int waves_per_cu, const char* tag) {
    const int iters = 2000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe<VARIANT>, dim3(blocks), dim3(64 * waves_per_cu), 0, 0, in, out, ticks, iters);
        hipDeviceSynchronize();
    }
    hipMemcpy(host, ticks, sizeof(long long) * blocks * waves_per_cu, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks * waves_per_cu; ++i) s += (double)host[i];
    printf("%-28s waves/SIMD=%d: %6.1f ticks per 32 exp2 per wave (+32 fma +32 add)\n", tag, waves_per_cu / 4,
           s / (blocks * waves_per_cu) / iters);
}

int main() {
    float *in, *out;
    long long* ticks;
    hipMalloc(&in, 4096);
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    hipMalloc(&ticks, sizeof(long long) * 256 * 16);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = -(float)(i % 37) * 0.37f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    long long* host = (long long*)malloc(sizeof(long long) * 256 * 16);
    for (int w = 4; w <= 12; w += 4) {
        run<0>(in, out, ticks, host, w, "v_exp_f32");
        run<1>(in, out, ticks, host, w, "polynomial (FMA pipe)");
        run<2>(in, out, ticks, host, w, "half and half");
    }
    return 0;
}
