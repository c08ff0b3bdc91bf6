// 3x3 convolution with the A operand REUSED from LDS across the three horizontal taps (round 6): the
// one-wave-per-SIMD persistent kernel of gemm_sw.hip (192 x 320 x 64 tiles, 96 x 160 wave tiles, accumulators in AGPRs, barrier
// inside the MFMA stream, direct 16-byte stores) with a different A path for stride-1 "same" convolutions whose image width is
// 16 / 32 / 64 (a 192-row tile is then a whole number RT of image rows):
//
//   * K order (dy, channel slice, dx) instead of (tap, slice): the three dx taps of one (dy, slice) are consecutive K-tiles.
//   * Per (dy, slice) ONE patch goes to LDS: the tile's RT image rows shifted by dy - 1, each with one pad pixel either side, pitch
//     WI + 8 pixels (a multiple of 8, so the 16-byte-chunk XOR swizzle by (pixel row & 7) of a 16-pixel fragment does not depend on
//     the image row), 27-36 KiB.  The dx tap is an ADDRESS OFFSET of the fragment reads (+ dx pixel rows = + dx * 128 bytes, swizzle
//     term (l15 + dx) & 7); nothing is re-fetched.  The A stream per K-tile drops from 24 KiB to 9-12 KiB (patch / 3), the whole
//     L2 -> LDS operand stream from 64 KiB to 49-52 KiB (-20..-23 %).
//   * Two patch slots + two W slots (40 KiB each): 137-154 KiB of LDS.  The next patch's LDS-DMA pieces (7-9 per wave) ride in the three
//     bodies of the current one, ahead of each body's ten W pieces.
//   * A full nine-tap halo (RT + 2 rows) does not fit twice next to the W slots (2 x 46 + 80 KiB); what the missing vertical reuse is
//     worth was priced on the linear kernel (tools/gemm_sw_ko.py, "A pieces every 3rd K-tile").
//
// Arithmetic: the same MFMA steps as the tap-gather kernels in a different K order -> results differ from theirs in the last fp32
// bits (tests: fp32 references at the kernel tolerance, the tap-gather result at 5e-4); a launch's K order is a function of its
// geometry and its HINTED row count only (dispatch), so the two- and three-branch engines of one model still agree bit for bit.
//
// Replaces: i2vgen-xl/pnp_utils.py conv1 / conv2 :78,:107 and the diffusers-0.26.3 ResnetBlock2D convolutions behind
// pipeline_i2vgen_xl.py:1146.
#include <type_traits>

#include "gemm_common.h"

namespace {

constexpr int SWH_MF = 6, SWH_BM = 192, SWH_BN = 320, SWH_B_BYTES = SWH_BN * 128;

__device__ __forceinline__ h8 swh_frag(unsigned addr) {
    h8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ h8 swh_frag_off(unsigned base, int off) {  // off: a constant after unrolling (16-bit immediate)
    h8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(off) : "memory");
    return v;
}
__device__ __forceinline__ void swh_lgkm(int n) {  // n is a constant after unrolling; the switch folds to one s_waitcnt
    switch (n) {
#define AV_LGW(k) case k: asm volatile("s_waitcnt lgkmcnt(" #k ")" ::: "memory"); break;
        AV_LGW(0) AV_LGW(1) AV_LGW(2) AV_LGW(3) AV_LGW(4) AV_LGW(5) AV_LGW(6) AV_LGW(7) AV_LGW(8) AV_LGW(9) AV_LGW(10)
        AV_LGW(11) AV_LGW(12) AV_LGW(13) AV_LGW(14) AV_LGW(15)
#undef AV_LGW
        default: asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); break;
    }
}
__device__ __forceinline__ float swh_acc(const float& a) {   // AGPR -> VGPR at the use (see gemm_sw.hip, sw_acc)
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}

}  // namespace

// WI: image width (16 / 32 / 64); EPI: 0 = bias (+ temb row vector), 1 = bias + residual
template <int WI, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_swh_kernel(const GemmK p) {
    constexpr int MF = SWH_MF, BM = SWH_BM, BN = SWH_BN, B_BYTES = SWH_B_BYTES;
    constexpr int RT = BM / WI;                 // image rows per tile
    constexpr int P = WI + 8;                   // patch pitch in pixels (pad pixel | WI pixels | pad pixel | 6 unused)
    constexpr int PPIX = RT * P;                // pixel rows of a patch (128 bytes each: one 64-channel slice)
    constexpr int NPP = (PPIX / 8 + 3) / 4;     // LDS-DMA pieces (8 pixel rows = 1 KiB) per wave and patch
    constexpr int PSLOT = NPP * 4096;           // bytes of a patch slot (whole pieces)
    constexpr int W_BASE = 2 * PSLOT;
    static_assert(BM % WI == 0 && PPIX % 8 == 0, "a tile is a whole number of image rows");
    __shared__ __attribute__((aligned(16))) char smem[2 * PSLOT + 2 * B_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int G = gridDim.x;
    const int b0 = ((G & 7) == 0) ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;   // XCD-contiguous tile order
    const int tilesM = (p.M + BM - 1) / BM;
    const int ntiles = tilesM * p.tilesN;
    const int ntap = p.nt0 + p.nt1;   // 64-channel slices of the (two-source) input
    const int nq = 3 * ntap;          // patches per tile: (dy, slice)
    const int Hi = p.Hi, HWi = p.Hi * p.Wi;
    const half_t* const zeros = p.zeros;

    // ---- patch stream (one patch ahead of the consumer): piece j of this wave = pixel rows (j * 4 + w) * 8 .. + 7 of the patch ----
    const int prow = lane >> 3;                 // pixel row inside the piece; also (patch pixel row & 7)
    const int hkc = (lane & 7) ^ prow;          // logical 16-byte chunk this lane fetches (source-side swizzle)
    int pk[NPP], py[NPP];                       // per piece: source pixel of the CENTRE row (dy = 1) or -1; output image row y
    int h_tile, h_dy = 0, h_slice = 0;          // patch stream position: tile, (dy, slice) of the patch being requested
    // source of the patch being requested, resolved when the stream moves -- NOT inside a K-tile body: which of the two inputs a slice
    // comes from selects between kernel ARGUMENTS, i.e. a scalar load plus an lgkmcnt(0) that drains the body's fragment-read pipeline
    const half_t* h_base;   // (source, slice, this lane's chunk)
    int h_ld, h_dyoff;      // leading dimension of that source; (dy - 1) * WI
    auto h_source = [&]() {
        const ASrc s = a_source(p, h_slice, hkc);
        h_base = s.base;
        h_ld = s.ld;
        h_dyoff = (h_dy - 1) * WI;
    };
    auto h_start = [&](int item) {
        h_tile = item;
        h_dy = h_slice = 0;
        h_source();
        if (item < ntiles) {
            const int mt = item / p.tilesN;
#pragma unroll
            for (int j = 0; j < NPP; ++j) {
                const int pr = (j * 4 + w) * 8 + prow;
                const int jr = pr / P, x = pr - jr * P - 1;
                const int m0 = mt * BM + jr * WI;                     // first output pixel of that image row
                const bool ok = pr < PPIX && x >= 0 && x < WI && m0 < p.M;
                const int img = m0 / HWi, y = (m0 - img * HWi) / WI;
                pk[j] = ok ? img * HWi + y * WI + x : -1;
                py[j] = y;
            }
        }
    };
    auto h_step = [&]() {
        if (h_tile >= ntiles) return;
        if (++h_slice == ntap) {
            h_slice = 0;
            if (++h_dy == 3) {
                h_start(h_tile + G);
                return;
            }
        }
        h_source();
    };
    auto h_piece = [&](int j, int slot) {   // j: constant after unrolling
        const int ys = py[j] + h_dy - 1;
        const bool ok = h_tile < ntiles && pk[j] >= 0 && (unsigned)ys < (unsigned)Hi;
        const half_t* src = h_base + (long long)(pk[j] + h_dyoff) * h_ld;
        glds16(ok ? src : zeros, smem + slot * PSLOT + (j * 4 + w) * 1024);
    };

    // ---- W stream (one K-tile ahead): K-tile (dy, slice, dx) reads W columns ((3 dy + dx) ntap + slice) * 64 ----
    const int srow0 = tid >> 3, kc = (tid & 7) ^ (srow0 & 7);
    const int wperm = 8 * ((srow0 & 15) >> 2) + 4 * (srow0 >> 4) + (srow0 & 3);   // W row permutation of gemm_sw.hip (plain form)
    const half_t* wbase;   // W row (n_blk + wperm), column kc * 8
    int w_tile, w_dy = 0, w_slice = 0, w_dx = 0;
    auto w_start = [&](int item) {
        w_tile = item;
        w_dy = w_slice = w_dx = 0;
        if (item < ntiles) {
            const int nt = item % p.tilesN;
            wbase = p.W + (size_t)(nt * BN + wperm) * p.Ktot + kc * 8;
        }
    };
    auto w_step = [&]() {
        if (w_tile >= ntiles) return;
        if (++w_dx == 3) {
            w_dx = 0;
            if (++w_slice == ntap) {
                w_slice = 0;
                if (++w_dy == 3) w_start(w_tile + G);
            }
        }
    };
    const size_t wrow32 = (size_t)32 * p.Ktot;   // 32 W rows in halves
    auto w_piece = [&](int j, int slot) {   // j: constant after unrolling
        const half_t* src = wbase + j * wrow32 + ((3 * w_dy + w_dx) * ntap + w_slice) * 64;
        glds16(w_tile < ntiles ? src : zeros, smem + W_BASE + slot * B_BYTES + (j * 256 + w * 64) * 16);
    };

    // ---- consumer ----
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned sm0 = (unsigned)(size_t)smem;
    const unsigned b_off = (unsigned)(W_BASE + (wc * 160 + l15) * 128);
    const unsigned wc0 = (unsigned)(((0 * 4 + lq) ^ (l15 & 7)) * 16), wc1 = (unsigned)(((1 * 4 + lq) ^ (l15 & 7)) * 16);
    // lane part of an A fragment address for tap dx, K-step ks: pixel row l15 + dx, chunk (ks * 4 + lq) ^ ((l15 + dx) & 7)
    auto a_lane = [&](int dx, int ks) { return (unsigned)((l15 + dx) * 128 + (((ks * 4 + lq) ^ ((l15 + dx) & 7)) * 16)); };
    // fragment mf of this wave: tile rows r0 .. r0 + 15 = image row r0 / WI of the tile, pixels r0 % WI ..; patch pixel row (r0 / WI) P + r0 % WI
    unsigned offA[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int r0 = wr * (MF * 16) + mf * 16;
        offA[mf] = (unsigned)(((r0 / WI) * P + (r0 % WI)) * 128);
    }

    f4 acc[MF][10];
    h8 a0[MF], wq[4];   // carried across K-tiles: A fragments of K-step 0 and the first three W fragments of the coming K-tile

    // One K-tile (tap dx of the patch in slot sp, W slot sw): the 20 MFMA groups of gemm_sw.hip's ktile.  Fillers behind group g: W
    // fragment g + 3, A fragments of K-step 1 behind groups 3..8, LDS-DMA pieces (this body's share of the NEXT patch first, then the
    // ten W pieces of the next K-tile: two per group behind groups 0..3, one per group after), and behind group 17 the wait / barrier
    // and the first fragments of the next K-tile: tap dx + 1 of the same patch, or tap 0 of the other slot.
    constexpr int HN0 = (NPP + 2) / 3, HN1 = HN0 + (NPP - HN0 + 1) / 2;   // patch pieces [0, HN0) ride in the dx = 0 body, [HN0, HN1) in dx = 1, rest in dx = 2
    auto ktile = [&](auto first_tag, auto dx_tag, int sp, int sw) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int DX = decltype(dx_tag)::value;
        constexpr int H0 = DX == 0 ? 0 : (DX == 1 ? HN0 : HN1), H1 = DX == 0 ? HN0 : (DX == 1 ? HN1 : NPP);
        constexpr int NPIECE = (H1 - H0) + 10;
        const unsigned sba = sm0 + sp * PSLOT, sbw = sm0 + sw * B_BYTES;
        const unsigned ab1 = sba + a_lane(DX, 1);
        const unsigned bbase[2] = {sbw + b_off + wc0, sbw + b_off + wc1};
        const unsigned nab = (DX == 2 ? sm0 + (sp ^ 1) * PSLOT : sba) + a_lane(DX == 2 ? 0 : DX + 1, 0);
        const unsigned nbbase = sm0 + (sw ^ 1) * B_BYTES + b_off + wc0;
        h8 a1[MF];
        int seq = 0, a1_seq = 0, w_seq[23] = {};
        int npiece = 0;
        auto piece = [&](int i) {   // i: constant after unrolling
            if (i < H1 - H0)
                h_piece(H0 + i, sp ^ 1);
            else
                w_piece(i - (H1 - H0), sw ^ 1);
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 20; ++g) {
            const int ks = g / 10, nf = g % 10;
            {
                int need = w_seq[g];
                if (g == 10 && a1_seq > need) need = a1_seq;
                if (need > 0 && g < 18) swh_lgkm(seq - need);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const h8& af = ks == 0 ? a0[mf] : a1[mf];
                if (FIRST && ks == 0)
                    acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[g & 3], af, (f4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                else
                    acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[g & 3], af, acc[mf][nf], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g >= 3 && g < 3 + MF) {
                a1[g - 3] = swh_frag(ab1 + offA[g - 3]);
                a1_seq = ++seq;
            }
            if (g + 3 < 20) {
                const int t = g + 3;
                wq[t & 3] = swh_frag_off(bbase[t / 10], (t % 10) * 2048);
                w_seq[t] = ++seq;
            }
            if (g < 4) {
                piece(npiece++);
                piece(npiece++);
            } else if (npiece < NPIECE) {
                piece(npiece++);
            }
            if (g == 17) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) a0[mf] = swh_frag(nab + offA[mf]);
                wq[0] = swh_frag_off(nbbase, 0);
                wq[1] = swh_frag_off(nbbase, 2048);
            }
            if (g == 18) wq[2] = swh_frag_off(nbbase, 2 * 2048);
            __builtin_amdgcn_sched_barrier(0);
        }
        static_assert(NPIECE <= 8 + 13, "all pieces of a body are issued by group 16");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    int tile = b0;
    if (tile >= ntiles) return;
    // prologue: patch 0 -> patch slot 0, W of K-tile 0 -> W slot 0
    h_start(tile);
    w_start(tile);
#pragma unroll
    for (int j = 0; j < NPP; ++j) h_piece(j, 0);
    h_step();
#pragma unroll
    for (int j = 0; j < 10; ++j) w_piece(j, 0);
    w_step();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
        const unsigned na = sm0 + a_lane(0, 0), nb = sm0 + b_off + wc0;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) a0[mf] = swh_frag(na + offA[mf]);
        wq[0] = swh_frag_off(nb, 0);
        wq[1] = swh_frag_off(nb, 2048);
        wq[2] = swh_frag_off(nb, 2 * 2048);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    int sp = 0, sw = 0;
    using T = std::true_type;
    using F = std::false_type;
    using D0 = std::integral_constant<int, 0>;
    using D1 = std::integral_constant<int, 1>;
    using D2 = std::integral_constant<int, 2>;

    while (true) {
        const int mt = tile / p.tilesN, nt = tile - mt * p.tilesN;
        const bool has_next = tile + G < ntiles;
        // (the first patch is peeled: its dx = 0 body starts the accumulators from the constant 0)
        ktile(T{}, D0{}, sp, sw);
        w_step();
        sw ^= 1;
        ktile(F{}, D1{}, sp, sw);
        w_step();
        sw ^= 1;
        ktile(F{}, D2{}, sp, sw);
        w_step();
        sw ^= 1;
        h_step();
        sp ^= 1;
        for (int q = 1; q < nq; ++q) {
            ktile(F{}, D0{}, sp, sw);
            w_step();
            sw ^= 1;
            ktile(F{}, D1{}, sp, sw);
            w_step();
            sw ^= 1;
            ktile(F{}, D2{}, sp, sw);
            w_step();
            sw ^= 1;
            h_step();
            sp ^= 1;
        }

        // ---------------- epilogue: as gemm_sw_kernel's (EPI 0 / 1) ----------------
        const int m_wave = mt * BM + wr * MF * 16;
        const int n_wave = nt * BN + wc * 160;
        h8 bias8[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) bias8[q] = *(const h8*)(p.bias != nullptr ? p.bias + n_wave + 32 * q + 8 * lq : p.zeros);
        h8 rr[EPI == 1 ? 3 : 1][5];
        auto load_res = [&](int mf) {   // mf: constant after unrolling
            if constexpr (EPI == 1) {
                int m = m_wave + mf * 16 + l15;
                m = m < p.M ? m : p.M - 1;
#pragma unroll
                for (int q = 0; q < 5; ++q) rr[mf % 3][q] = *(const h8*)(p.R + (size_t)m * p.ldr + n_wave + 32 * q + 8 * lq);
            }
        };
        load_res(0);
        load_res(1);
        auto rows = [&](auto rv_tag) {
            constexpr bool HAS_RV = decltype(rv_tag)::value;
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int m = m_wave + mf * 16 + l15;
                const bool ok = m < p.M;
                half_t* crow = p.C + (size_t)m * p.ldc + n_wave + 8 * lq;
                if (mf + 2 < MF) load_res(mf + 2);
                h8 tv[HAS_RV ? 5 : 1];
                if constexpr (HAS_RV) {
                    const half_t* rv = p.rowvec + (size_t)((ok ? m : 0) / p.rowvec_div) * p.ldrv + n_wave + 8 * lq;
#pragma unroll
                    for (int q = 0; q < 5; ++q) tv[q] = *(const h8*)(rv + 32 * q);
                }
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    h8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v0 = swh_acc(acc[mf][2 * q][e]) + (float)bias8[q][e], v1 = swh_acc(acc[mf][2 * q + 1][e]) + (float)bias8[q][4 + e];
                        if constexpr (HAS_RV) {
                            v0 += (float)tv[q][e];
                            v1 += (float)tv[q][4 + e];
                        }
                        o[e] = (half_t)v0;
                        o[4 + e] = (half_t)v1;
                    }
                    if constexpr (EPI == 1) o = o + rr[mf % 3][q];
                    if (ok) *(h8*)(crow + 32 * q) = o;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (EPI == 1) {
            rows(F{});
        } else {
            if (p.rowvec != nullptr)
                rows(T{});
            else
                rows(F{});
        }
        if (!has_next) break;
        tile += G;
    }
}

// ---------------------------------------------------------------------------------------------------------
bool av_gemm_swh_eligible(const AnyV2VGemmDesc* d) {
    return d->mode == MODE_CONV2D && d->stride == 1 && d->up == 0 && d->asym == 0 && d->Hi == d->Ho && d->Wi == d->Wo &&
           (d->Wi == 16 || d->Wi == 32 || d->Wi == 64) && d->N % 320 == 0 && d->act == ACT_NONE && d->C0 % 64 == 0 && d->C1 % 64 == 0 &&
           !(d->R != nullptr && d->rowvec != nullptr) && d->ldc % 8 == 0 && (d->R == nullptr || d->ldr % 8 == 0) &&
           (d->rowvec == nullptr || d->ldrv % 8 == 0) && (long long)d->M * 1 < (1ll << 31);
}

template <int WI>
static void swh_launch_w(const GemmK& k, const AnyV2VGemmDesc* d, dim3 grid, hipStream_t s) {
    if (d->R != nullptr)
        hipLaunchKernelGGL((gemm_swh_kernel<WI, 1>), grid, dim3(256), 0, s, k);
    else
        hipLaunchKernelGGL((gemm_swh_kernel<WI, 0>), grid, dim3(256), 0, s, k);
}

int av_gemm_swh_launch(GemmK& k, const AnyV2VGemmDesc* d, hipStream_t s) {
    const int tiles = ((d->M + SWH_BM - 1) / SWH_BM) * (d->N / 320);
    k.tilesN = d->N / 320;
    const dim3 grid(tiles < 256 ? tiles : 256);
    if (d->Wi == 64)
        swh_launch_w<64>(k, d, grid, s);
    else if (d->Wi == 32)
        swh_launch_w<32>(k, d, grid, s);
    else
        swh_launch_w<16>(k, d, grid, s);
    return av_launch_status("gemm_swh");
}
