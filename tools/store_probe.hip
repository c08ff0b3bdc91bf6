// Microbenchmark (gfx950): sustained global-store rate per CU for 16-byte-per-lane stores (1 KiB per wave-instruction),
// as issued by the GEMM epilogues.  Variants: waves per CU, row-contiguous 320-byte segments (epilogue pattern) vs
// fully contiguous 1-KiB pieces, plain vs non-temporal.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_probe tools/store_probe.hip && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int PATTERN, bool NT>
__global__ void probe(_Float16* dst, long long* out, int iters, size_t bytes_per_block) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    char* base = (char*)dst + (size_t)blockIdx.x * bytes_per_block;
    h8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)(float)(lane + e);
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        char* p;
        if (PATTERN == 0) {  // contiguous: wave writes 1 KiB, waves back to back
            p = base + ((size_t)(it * nw + w) * 64 + lane) * 16;
        } else if (PATTERN == 1) {  // 16 rows x 320 B per 5 instructions, row pitch 640 B, other half never written
            const int c = (it % 5) * 64 + lane, row = c / 20, cc = c % 20;
            p = base + ((size_t)((it / 5) * nw + w) * 16 + row) * 640 + cc * 16;
        } else {             // the GEMM epilogue: wave pairs write the two 320-byte halves of the same 16 rows
            const int c = (it % 5) * 64 + lane, row = c / 20, cc = c % 20;
            p = base + ((size_t)((it / 5) * (nw / 2) + (w >> 1)) * 16 + row) * 640 + (w & 1) * 320 + cc * 16;
        }
        if (NT)
            __builtin_nontemporal_store(v, (h8*)p);
        else
            *(h8*)p = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * nw + w] = t1 - t0;
}

template <int PATTERN, bool NT>
static void run(_Float16* dst, long long* out, long long* host, int waves, const char* tag, int blocks = 256) {
    const int iters = 600;
    const size_t bpb = (size_t)iters * waves * 1024 * 2;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<PATTERN, NT>), dim3(blocks), dim3(waves * 64), 0, 0, dst, out, iters, bpb);
        hipDeviceSynchronize();
    }
    hipMemcpy(host, out, sizeof(long long) * blocks * waves, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks * waves; ++i) s += (double)host[i];
    const double ticks = s / (blocks * waves);
    printf("%-36s CUs=%3d waves/CU=%2d: %8.0f ticks for %d KiB per CU -> %5.1f B/tick/CU\n", tag, blocks, waves, ticks, iters * waves,
           iters * waves * 1024.0 / ticks);
}

int main() {
    _Float16* dst;
    long long* out;
    const size_t total = (size_t)256 * 600 * 16 * 1024 * 2 + (1 << 20);
    hipMalloc(&dst, total);
    hipMalloc(&out, sizeof(long long) * 256 * 16);
    long long* host = (long long*)malloc(sizeof(long long) * 256 * 16);
    for (int waves = 4; waves <= 16; waves *= 2) {
        run<0, false>(dst, out, host, waves, "contiguous 1 KiB pieces");
        run<1, false>(dst, out, host, waves, "320-byte segments, half rows only");
        run<2, false>(dst, out, host, waves, "320-byte halves by wave pairs");
        run<0, true>(dst, out, host, waves, "contiguous, non-temporal");
    }
    // how much of that is the CU's own write path and how much the HBM shared by all CUs: fewer CUs storing at once
    for (int blocks = 8; blocks <= 256; blocks *= 2) {
        run<0, false>(dst, out, host, 8, "contiguous 1 KiB pieces", blocks);
        run<2, false>(dst, out, host, 8, "320-byte halves by wave pairs", blocks);
    }
    return 0;
}
