// Reproducer (compile only: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only tools/wait_probe.hip -o - | grep -n "ds_read\|s_waitcnt\|mfma"):
// with an LDS-DMA load (global_load_lds) in flight, hipcc (ROCm 7.2) waits lgkmcnt(0) before every use of a ds_read result, also
// when a younger ds_read is outstanding (lgkmcnt(1) would do: LDS reads return in order, the DMA completes on vmcnt).  Remove the
// glds16 call inside the loop and the same code gets lgkmcnt(1).  gemm.hip's persistent kernel therefore issues its fragment
// reads as inline asm with hand-counted waits (profiles/r02_gemm_frag_wait_ab.txt).
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__global__ void k(float* out, const h8* in, int n) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int w = threadIdx.x >> 6;
    f4 acc = {0,0,0,0};
    int stage = 0;
    glds16(in + threadIdx.x, smem + w * 1024);
    for (int it = 0; it < n; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const h8* s = (const h8*)(smem + stage * 32768);
        char* st = smem + (stage ^ 1) * 32768;
        h8 a = s[threadIdx.x ^ 1];
        h8 b = s[(threadIdx.x ^ 2) + 64];
        __builtin_amdgcn_sched_barrier(0);
        h8 c = s[(threadIdx.x ^ 3) + 128];
        glds16(in + it * 512 + threadIdx.x, st + w * 1024);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        h8 d = s[(threadIdx.x ^ 4) + 192];
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, c, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        h8 e = s[(threadIdx.x ^ 5) + 256];
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, d, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, e, acc, 0, 0, 0);
        stage ^= 1;
    }
    *(f4*)(out + threadIdx.x*4) = acc;
}
