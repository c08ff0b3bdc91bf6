// Per-CU ingest rate of activation rows ([T][320] fp16, 640-byte rows) into REGISTERS for the weight-stationary GEMM's geometry:
// 256 blocks x 8 waves, a wave walks 32-row strips (20 KB) of its block's row range, S blocks (same XCD) read the same range at
// the same time (L2 hits for S - 1 of them).  Which lane -> address pattern and how many loads in flight reach what rate?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/ingest_probe tools/probes/ingest_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// PAT 0: fragment-shaped (row l&15, chunk l>>4; 16 rows x 64 B of one 32-k step)   1: quad-coalesced (row l>>2, chunk l&3)
// PAT 2: full lines (row l>>3, chunk l&7: 8 rows x 128 B)                          3: flat (the strip as 20 contiguous KB)
template <int PAT, int DEPTH>
__global__ __launch_bounds__(512) void ingest(const char* A, int S, int px, int nstrips, int spr, unsigned* sink) {
    __shared__ char big[120 * 1024];
    big[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (idx >= px * S) return;
    const int range = xcd * px + idx / S;
    const int s_begin = range * spr, s_end = min(s_begin + spr, nstrips);
    u4 acc = {0, 0, 0, 0};
    for (int strip = s_begin + w; strip < s_end; strip += 8) {
        const char* base = A + (size_t)strip * 32 * 640;
#pragma unroll
        for (int g = 0; g < 20 / DEPTH; ++g) {
            u4 v[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                const int j = g * DEPTH + i;   // load index 0..19 within the strip
                const char* p;
                if (PAT == 0) p = base + (size_t)((j & 1) * 16 + (lane & 15)) * 640 + (j >> 1) * 64 + (lane >> 4) * 16;
                if (PAT == 1) p = base + (size_t)((j & 1) * 16 + (lane >> 2)) * 640 + (j >> 1) * 64 + (lane & 3) * 16;
                if (PAT == 2) p = base + (size_t)((j & 3) * 8 + (lane >> 3)) * 640 + (j >> 2) * 128 + (lane & 7) * 16;
                if (PAT == 3) p = base + (size_t)j * 1024 + lane * 16;
                v[i] = *(const u4*)p;
            }
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) acc ^= v[i];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567) sink[0] = 1;
}

template <int PAT, int DEPTH>
static void run(const char* A, int S, int M, unsigned* sink, const char* name) {
    const int px = 32 / S, nstrips = M / 32, nranges = 8 * px, spr = (nstrips + nranges - 1) / nranges;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        ingest<PAT, DEPTH><<<256, 512>>>(A, S, px, nstrips, spr, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes_cu = (double)M * 640 * S / (8.0 * px * S);   // per active CU
    printf("S=%2d %-14s depth %2d: %7.1f us  per-CU ingest %6.1f GB/s  (chip L1-side %5.2f TB/s, unique %5.2f TB/s)\n", S, name, DEPTH,
           best * 1e3, bytes_cu / best / 1e6, (double)M * 640 * S / best / 1e9, (double)M * 640 / best / 1e9);
}

int main() {
    const int M = 196608;
    char* A;
    unsigned* sink;
    hipMalloc(&A, (size_t)M * 640 + 4096);
    hipMalloc(&sink, 4);
    hipMemset(A, 1, (size_t)M * 640);
    for (int S : {1, 2, 16}) {
        run<0, 20>(A, S, M, sink, "fragment");
        run<1, 20>(A, S, M, sink, "quad 64B");
        run<2, 20>(A, S, M, sink, "full lines");
        run<3, 20>(A, S, M, sink, "flat");
        run<1, 10>(A, S, M, sink, "quad 64B");
        run<1, 4>(A, S, M, sink, "quad 64B");
        run<2, 10>(A, S, M, sink, "full lines");
        run<2, 4>(A, S, M, sink, "full lines");
        run<3, 4>(A, S, M, sink, "flat");
    }
    return 0;
}
