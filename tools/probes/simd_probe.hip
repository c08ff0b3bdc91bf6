// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id [15:13])
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/simd_probe tools/probes/simd_probe.hip && /tmp/simd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d;
    const int nb = 512;
    hipMalloc(&d, nb * 8 * 4);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 0, 0, d);
    unsigned h[nb * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int pair_w4 = 0, pair_w1 = 0, other = 0;
    for (int b = 0; b < nb; ++b) {
        int simd[8];
        for (int w = 0; w < 8; ++w) simd[w] = (h[b * 8 + w] >> 4) & 3;
        bool w4 = true, w1 = true;
        for (int w = 0; w < 4; ++w) w4 &= simd[w] == simd[w + 4];
        for (int w = 0; w < 8; w += 2) w1 &= simd[w] == simd[w + 1];
        pair_w4 += w4; pair_w1 += w1; other += !(w4 || w1);
        if (b < 6 || b == 300) {
            printf("block %3d: simd of waves 0..7 =", b);
            for (int w = 0; w < 8; ++w) printf(" %d", simd[w]);
            printf("   wave_id:");
            for (int w = 0; w < 8; ++w) printf(" %d", h[b * 8 + w] & 15);
            printf("   cu %d se %d\n", (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7);
        }
    }
    printf("blocks whose SIMD partners are (w, w+4): %d, (w, w+1): %d, neither: %d of %d\n", pair_w4, pair_w1, other, nb);
    return 0;
}
