// Which XCD does block b of a 256 x 512-thread, LDS-heavy (1 block per CU) launch run on?  And: do blocks that stream the
// SAME buffer at the same time get it from their XCD's L2?  (input for gemm_ws.hip's block -> (xcd, slab, range) mapping)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcc_probe tools/probes/xcc_probe.hip && /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(512) void who(int* out) {
    __shared__ char big[120 * 1024];
    big[threadIdx.x] = 0;
    if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20);
}

// every block streams `bytes` of `buf` starting at (group * bytes), group = f(blockIdx); 16-byte loads, coalesced
template <int MAP>
__global__ __launch_bounds__(512) void stream(const uint4* buf, size_t vecs_per_group, int groups, unsigned* sink) {
    __shared__ char big[120 * 1024];
    big[threadIdx.x] = 0;
    int g;
    if (MAP == 0) g = (blockIdx.x & 7) * (groups / 8) + (blockIdx.x >> 3) % (groups / 8);   // same-XCD blocks share a group
    else g = blockIdx.x % groups;                                                            // sharers spread over XCDs
    const uint4* p = buf + (size_t)g * vecs_per_group;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < vecs_per_group; i += 512) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345) sink[0] = acc;
}

int main() {
    int* d;
    hipMalloc(&d, 256 * 4);
    who<<<256, 512>>>(d);
    std::vector<int> h(256);
    hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost);
    int agree = 0;
    for (int b = 0; b < 256; ++b) agree += (h[b] == (b & 7));
    printf("blocks with xcc_id == blockIdx %% 8: %d / 256\nfirst 32:", agree);
    for (int b = 0; b < 32; ++b) printf(" %d", h[b]);
    printf("\n");
    const size_t total = 1024ull << 20;  // 256 groups x 4 MB
    uint4* buf;
    unsigned* sink;
    hipMalloc(&buf, total);
    hipMalloc(&sink, 4);
    hipMemset(buf, 1, total);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int groups : {256, 128, 64, 32, 16, 8}) {          // sharers per group = 256 / groups
        for (size_t mb_per_group : {1, 4}) {
            const size_t vecs = (mb_per_group << 20) / 16;
            for (int map = 0; map < 2; ++map) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipEventRecord(e0);
                    if (map == 0) stream<0><<<256, 512>>>(buf, vecs, groups, sink);
                    else stream<1><<<256, 512>>>(buf, vecs, groups, sink);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                const double read = 256.0 * (mb_per_group << 20), uniq = (double)groups * (mb_per_group << 20);
                printf("groups %3d (sharers %3d) x %zu MB, map %s: %.1f us, L1-side %.2f TB/s, unique bytes %.2f TB/s\n", groups,
                       256 / groups, mb_per_group, map == 0 ? "same-XCD" : "spread  ", best * 1e3, read / best / 1e9, uniq / best / 1e9);
            }
        }
    }
    return 0;
}
