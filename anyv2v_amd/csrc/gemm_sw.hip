// One-wave-per-SIMD persistent gather-GEMM for gfx950 (round 6): the same operations as gemm_big_kernel (Linear / conv2d 3x3 /
// temporal (3,1,1) conv over channels-last token matrices, fp16 in, fp32 MFMA accumulate, fp16 out; bias / temb row vector /
// residual / GEGLU epilogues) on the same 192 x 320 x 64 block tile, with FOUR waves instead of eight:
//
//   * 256 threads = one wave per SIMD, __launch_bounds__(256, 1): the wave owns the whole 512-entry register file of its SIMD --
//     240 accumulators (wave tile 96 x 160 = 6 x 10 fragments of v_mfma_f32_16x16x32_f16) in AGPRs, fragments / addresses /
//     epilogue operands in the 256 arch VGPRs.  A 96 x 160 wave tile reads 16 fragments per 60 MFMAs (0.27 ds_read_b128 per MFMA;
//     the eight-wave kernel's 48 x 160 tile: 0.43), and nothing on the SIMD competes with the wave for issue slots.
//   * The K-tile stream is seamless across K-tiles AND output tiles: the block barrier sits after group 17 of the K-tile's 20 MFMA
//     groups (by then every fragment of the tile has been read), the first fragments of the next K-tile are requested right behind
//     it and land under the last 12 MFMAs.  LDS-DMA pieces of the next K-tile ride in the first groups, one or two per group.
//   * Output leaves the registers directly: the LDS-DMA source row of every W piece is permuted so that a lane's accumulators of a
//     fragment PAIR are 8 consecutive output channels -> one 16-byte global store per (16-row fragment, 32-channel pair), no LDS
//     turn, no block barrier in the epilogue; the stores drain under the next tile's first K-tile.
//
// Arithmetic is the eight-wave kernel's, element for element (K ascending in 32-deep MFMA steps from a zero accumulator, then
// + bias (+ temb) in fp32, one rounding to fp16, residual added in fp16): results are bit-equal to gemm_big_kernel / gemm_mfma_kernel
// (tests/gpu_checks.py::check_gemm_big(extra = bit21)).
//
// Replaces (reference = TIGER-AI-Lab/AnyV2V): i2vgen-xl/pnp_utils.py conv1/conv2 :78,:107, conv_shortcut :117-122, residual :124,
// attn.to_q/to_k/to_v :175,:182-183, attn.to_out[0] :216, and the diffusers-0.26.3 FeedForward (GEGLU up / down projections) and
// TemporalConvLayer behind pipeline_i2vgen_xl.py:1146.
#include <stdlib.h>
#include <type_traits>

#include "gemm_common.h"

namespace {

#ifndef SW_MF_
#define SW_MF_ 6
#endif
constexpr int SW_MF = SW_MF_;          // 16-row fragments per wave (6: 96 rows, two wave rows -> 192-row block tile)
constexpr int SW_BM = 32 * SW_MF, SW_BN = 320;
constexpr int SW_A_BYTES = SW_BM * 128, SW_B_BYTES = SW_BN * 128;   // one K-tile (64 deep): 24 KiB of A, 40 KiB of W
#ifndef SW_ALA
#define SW_ALA 1   // K-tiles the A stream runs ahead of the consumer: 1 = two A slots, 2 = a ring of three (see the kernel)
#endif

template <bool REAL = true>
__device__ __forceinline__ h8 sw_frag(unsigned base, int off) {  // off: a constant after unrolling (16-bit immediate)
    h8 v;
    if constexpr (REAL)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(off) : "memory");
    else
    {   // (probe: a register constant instead of the read)
        u4 x;
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("v_mov_b32 %0, %1" : "=v"(x[e]) : "v"(0x3c003c00u + base * 0));
        v = __builtin_bit_cast(h8, x);
    }
    return v;
}
__device__ __forceinline__ void sw_lgkm(int n) {  // n is a constant after unrolling; the switch folds to one s_waitcnt
    switch (n) {
#define AV_LGW(k) case k: asm volatile("s_waitcnt lgkmcnt(" #k ")" ::: "memory"); break;
        AV_LGW(0) AV_LGW(1) AV_LGW(2) AV_LGW(3) AV_LGW(4) AV_LGW(5) AV_LGW(6) AV_LGW(7) AV_LGW(8) AV_LGW(9) AV_LGW(10)
        AV_LGW(11) AV_LGW(12) AV_LGW(13) AV_LGW(14) AV_LGW(15)
#undef AV_LGW
        default: asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); break;
    }
}

// accumulator element -> VGPR, AT the use: left to hipcc, the AGPR -> VGPR copies of all 240 accumulators are hoisted to the top of the
// epilogue (they are copies of phi values), which overflows the 256 arch VGPRs into AGPRs and the accumulators into scratch
__device__ __forceinline__ float sw_acc(const float& a) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}

// gather addressing of the six A rows a thread stages per K-tile (rows srow0 + 32 i): inside one (tap, source) run consecutive
// K-tiles only advance the channel offset; the row -> shifted-row math is redone when the tap or the source changes
// ORD (conv2d): 0 = tap-major K order (tap, channel slice) as gemm_big_kernel; 1 = slice-major (channel slice, dy, dx): the three dx taps
// of one (slice, dy) are consecutive K-tiles and touch the same A lines shifted by one pixel -- L1 (TCP) hits when nothing else
// allocates there in between (the W pieces then go past L1, `sc1`)
template <int MODE, int ORD = 0>
struct SwGen {
    const half_t* ap[SW_MF];
    int astep[SW_MF];
    int ktc, tap;
    __device__ __forceinline__ void recompute(const GemmK& p, const RowInfo (&ri)[SW_MF], int kc) {
        const ASrc s = a_source(p, ktc, kc);
#pragma unroll
        for (int i = 0; i < SW_MF; ++i) {
            const int sr = src_row<MODE>(p, ri[i], tap);
            ap[i] = a_addr(p, s, sr);
            astep[i] = sr < 0 ? 0 : 64;
        }
    }
    __device__ __forceinline__ void start(const GemmK& p, const RowInfo (&ri)[SW_MF], int kc, int kt0 = 0, int ntap = 1) {
        tap = kt0 / ntap;          // (tap-major order; ORD 1 launches always start at K-tile 0)
        ktc = kt0 - tap * ntap;
        recompute(p, ri, kc);
    }
    __device__ __forceinline__ void next(const GemmK& p, const RowInfo (&ri)[SW_MF], int kc, int ntap) {
        if constexpr (ORD == 1) {
            if (++tap == p.taps) {
                tap = 0;
                ++ktc;
            }
            recompute(p, ri, kc);
            return;
        }
        if (++ktc == ntap) {
            ktc = 0;
            ++tap;
        }
        if (ktc == 0 || ktc == p.nt0) {
            recompute(p, ri, kc);
        } else {
#pragma unroll
            for (int i = 0; i < SW_MF; ++i) ap[i] += astep[i];
        }
    }
    // The same step in two parts (ORD 0): `bump` = the pointer adds, UNCONDITIONAL, issued inside the K-tile body as fillers between
    // MFMAs; `count` = the counters and -- when the tap or the source changes -- the recomputation that overwrites the bumped pointers,
    // between two bodies.  With one wave per SIMD every VALU instruction between two bodies is matrix-pipe idle time.
    __device__ __forceinline__ void bump() {
#pragma unroll
        for (int i = 0; i < SW_MF; ++i) ap[i] += astep[i];
    }
    __device__ __forceinline__ void count(const GemmK& p, const RowInfo (&ri)[SW_MF], int kc, int ntap) {
        if (++ktc == ntap) {
            ktc = 0;
            ++tap;
        }
        if (ktc == 0 || ktc == p.nt0) recompute(p, ri, kc);
    }
};

__device__ __forceinline__ void glds16_sc1(const half_t* g, char* lds_wave_base) {   // the same piece past L1 (sc1): L2-served, no TCP line
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 16);
}

// W row (relative to the wave slab's first row) that LDS row `s + 32 * piece` of the slab holds, as  row = wrow_thread(s) + wrow_piece(piece):
//  plain : LDS rows of a fragment pair (32 rows) hold W rows 8 (i / 4) + 4 f + i % 4  (f = fragment of the pair, i = row in it)
//  GEGLU : W comes as 32-row blocks [16 h | 16 gate]; two blocks form a 64-row group [hA gA hB gB] whose A / B fragments interleave
//          4-channel runs the same way; the fifth block of the slab stays in natural order (its outputs leave as 8-byte stores)
template <bool GEGLU>
__device__ __forceinline__ int sw_wrow_thread(int s) {
    const int i = s & 15, f = s >> 4;
    if constexpr (GEGLU)
        return 32 * (i >> 3) + 16 * f + 8 * ((i >> 2) & 1) + (i & 3);
    else
        return 8 * (i >> 2) + 4 * f + (i & 3);
}
template <bool GEGLU>
__device__ __forceinline__ constexpr int sw_wrow_piece(int pl) {   // pl: piece within the wave slab, 0..4
    if constexpr (GEGLU)
        return pl < 4 ? 64 * (pl >> 1) + 4 * (pl & 1) : 128;
    else
        return 32 * pl;
}

}  // namespace

// EPI: 0 = bias (+ temb row vector), 1 = bias + residual, 2 = GEGLU
// KO (probe builds, tools/gemm_sw_ko.py): 1 = the K loop issues no LDS-DMA piece, 2 = no fragment reads either, 3 = pieces but no MFMAs,
// 4 = W pieces only, 5 = A pieces from the zero line (issued, but always cache hits), 6 = A pieces in every third K-tile only;
// conv2d: 7 = slice-major K order (slice, dy, dx) with the W pieces past L1 (sc1), 8 = that order with plain W pieces
// SK (stream-K): the block owns a contiguous range of the launch's (tile, K-tile) units instead of whole tiles -- for launches whose
// tiles cannot fill 256 CUs for a whole number of rounds.  Tiles cut by a range boundary leave their raw fp32 accumulators in a
// workspace slab ([2 x blocks][192][320] fp32, columns in W-row order); gemm_sw_fixup_kernel sums a tile's slabs in K order and
// finishes them (bias / temb / GEGLU / residual).  Tiles a block covers alone take the normal epilogue.
template <int MODE, int EPI, int KO = 0, bool SK = false>
__global__ __launch_bounds__(256, 1) void gemm_sw_kernel(const GemmK p) {
    constexpr bool GEGLU = EPI == 2;
    constexpr int MF = SW_MF, BM = SW_BM, BN = SW_BN, A_BYTES = SW_A_BYTES, B_BYTES = SW_B_BYTES;
    // LDS: a ring of THREE A slots (24 KiB each) and two W slots (40 KiB each), 152 KiB.  The A operand is the one that misses
    // (every A line is a compulsory HBM / Infinity-Cache fetch for the first block of an XCD that touches it; W lines are L2 hits for
    // all but one block per XCD): with two stages a K-tile's A pieces have one K-tile (~1.5 us) to land, less than a loaded HBM miss,
    // and the K loop stalls on them every K-tile (tools/gemm_sw_ko.py: W pieces only = 1.42-1.51 PF, with A 0.85-1.2).  The third A slot
    // lets A run TWO K-tiles ahead: body g issues W(g + 1) first, then A(g + 2), and waits vmcnt(MF) -- all but the youngest MF
    // loads, i.e. A(g + 2), may still be in flight (loads retire in order; no store is ever issued inside a body).
    // Measured (profiles/r06_gemm_sw_ab_aring.txt): the ring is 7-10 % SLOWER than two A slots on every shape -- what the A pieces cost is
    // not their latency but their share of the L2 -> LDS fill rate (~22 B/clk per CU, tools/gemm_sw_ko.py) -- so SW_ALA = 1 ships.
    constexpr int ALA = SW_ALA, NA = ALA + 1;   // A look-ahead in K-tiles, A slots
    constexpr int W_BASE = NA * A_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[NA * A_BYTES + 2 * B_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int G = gridDim.x;
    // tile order: as gemm_big_kernel (classic = N-fastest in XCD-contiguous runs; rastered = rast_gm x rast_gn super-tiles per XCD round)
    const bool rast = !SK && p.rast_gm > 0;
    const int b0 = rast ? (int)blockIdx.x : (((G & 7) == 0) ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x);
    const int tilesM = (p.M + BM - 1) / BM;
    const int ntiles = rast ? G * ((p.rast_sm * p.rast_sn + 7) >> 3) : tilesM * p.tilesN;
    auto decode = [&](int t, int& mt, int& nt) -> bool {
        if (!rast) {
            mt = t / p.tilesN;
            nt = t - mt * p.tilesN;
            return true;
        }
        const int q = (t & 7) + 8 * (t / G), j = (t % G) >> 3;
        int sm, sn;
        if (p.rast_nfast) {
            sm = q / p.rast_sn;
            sn = q - sm * p.rast_sn;
        } else {
            sn = q / p.rast_sm;
            sm = q - sn * p.rast_sm;
        }
        const int jm = j / p.rast_gn, jn = j - jm * p.rast_gn;
        mt = sm * p.rast_gm + jm;
        nt = sn * p.rast_gn + jn;
        return q < p.rast_sm * p.rast_sn && mt < tilesM;
    };
    auto next_valid = [&](int t) {
        int mt_, nt_;
        while (t < ntiles && !decode(t, mt_, nt_)) t += G;
        return t;
    };

    // ---- producers: thread -> rows srow0 + 32 i of the A tile / of every 32-row W piece, physical chunk pc, logical chunk kc.
    // Two independent streams over the block's (tile, K-tile) sequence: A runs two K-tiles ahead of the consumer, W one.
    const int srow0 = tid >> 3, pc = tid & 7, kc = pc ^ (srow0 & 7);
    const int ntap = p.nt0 + p.nt1;
    const int nk = p.taps * ntap;
    RowInfo ri[MF];
    constexpr int ORD = (MODE == MODE_CONV2D && (KO == 7 || KO == 8)) ? 1 : 0;   // (probe builds: slice-major conv K order)
    constexpr bool WSC1 = KO == 7;                                                // (probe builds: W pieces past L1)
    SwGen<MODE, ORD> gen;
    int w_tap = 0, w_slice = 0;   // ORD 1: position of the W stream inside the tile
    int a_tile, a_kt;      // A stream position (a_tile >= ntiles: past the end, its pieces read the zero line)
    const half_t* bptr;    // W row of LDS row srow0 of wave slab 0 (pieces: + sw_wrow_piece rows, slab 1: + 160 rows)
    const half_t* bptr4;   // GEGLU: the natural-order fifth block
    int w_tile, w_kt;      // W stream position
    // SK: units [u0, u1) of the flattened (tile, K-tile) space; a_left / w_left = units of the range the streams have not requested yet
    const long long U = (long long)ntiles * nk;
    const int u0 = SK ? (int)(U * b0 / G) : 0, u1 = SK ? (int)(U * (b0 + 1) / G) : 0;
    int a_left = u1 - u0, w_left = u1 - u0;
    auto a_start = [&](int item, int kt0 = 0) {
        a_tile = item;
        a_kt = kt0;
        if (item < ntiles) {
            int mt, nt;
            decode(item, mt, nt);
#pragma unroll
            for (int i = 0; i < MF; ++i) ri[i] = make_row<MODE>(p, mt * BM + srow0 + 32 * i);
            gen.start(p, ri, kc, kt0, ntap);
        }
    };
    auto a_step = [&]() {
        if (a_tile >= ntiles) return;
        if constexpr (SK) {
            if (--a_left <= 0) {
                a_tile = ntiles;
                return;
            }
        }
        if (++a_kt == nk) {
            a_start(SK ? a_tile + 1 : next_valid(a_tile + G));
        } else {
            if constexpr (ORD == 0 && ALA == 1)
                gen.count(p, ri, kc, ntap);   // (its pointer adds were issued inside the body: stream_bump)
            else
                gen.next(p, ri, kc, ntap);
        }
    };
    auto w_start = [&](int item, int kt0 = 0) {
        w_tile = item;
        w_kt = kt0;
        if (item < ntiles) {
            int mt, nt;
            decode(item, mt, nt);
            bptr = p.W + (size_t)(nt * BN + sw_wrow_thread<GEGLU>(srow0)) * p.Ktot + kc * 8 + (size_t)kt0 * 64;
            bptr4 = p.W + (size_t)(nt * BN + srow0) * p.Ktot + kc * 8 + (size_t)kt0 * 64;
            w_tap = w_slice = 0;
        }
    };
    auto w_step = [&]() {
        if (w_tile >= ntiles) return;
        if constexpr (SK) {
            if (--w_left <= 0) {
                w_tile = ntiles;
                return;
            }
        }
        if (++w_kt == nk) {
            w_start(SK ? w_tile + 1 : next_valid(w_tile + G));
        } else {
            if constexpr (ORD == 1) {   // W column block of K-tile (slice, tap): tap * ntap + slice
                const int prev = w_tap * ntap + w_slice;
                if (++w_tap == p.taps) {
                    w_tap = 0;
                    ++w_slice;
                }
                bptr += (w_tap * ntap + w_slice - prev) * 64;
            } else if constexpr (ALA != 1) {
                bptr += 64;
                bptr4 += 64;
            }   // (ORD 0, ALA 1: bptr / bptr4 were advanced inside the body: stream_bump)
        }
    };
    // the unconditional part of both streams' step, issued inside the K-tile body (behind MFMA group 13; ALA 1 only): harmless when the
    // step then turns out to be a tile switch or a recomputation, which overwrite every pointer
    auto stream_bump = [&]() {
        if constexpr (ORD == 0 && ALA == 1) {
            gen.bump();
            bptr += 64;
            bptr4 += 64;
        }
    };
    auto a_piece = [&](int i, int slot, bool fetch) {   // i: constant after unrolling
        if constexpr (KO == 1 || KO == 2 || KO == 4) return;
        if constexpr (KO == 5) fetch = false;           // (probe: A pieces issued, all from the 256-byte zero line)
        if constexpr (KO == 6) {                        // (probe: A pieces in one K-tile of three -- what a 3x-reused patch would request)
            if (a_kt % 3 != 0) return;
        }
        glds16(fetch ? gen.ap[i] : p.zeros, smem + slot * A_BYTES + (i * 256 + w * 64) * 16);
    };
    auto w_piece = [&](int j, int slot, bool fetch) {   // j: constant after unrolling
        if constexpr (KO == 1 || KO == 2) return;
        const int ws = j / 5, pl = j % 5;
        const half_t* src = (GEGLU && pl == 4) ? bptr4 + (size_t)(ws * 160 + 128) * p.Ktot
                                               : bptr + (size_t)(ws * 160 + sw_wrow_piece<GEGLU>(pl)) * p.Ktot;
        if constexpr (WSC1)
            glds16_sc1(fetch ? src : p.zeros, smem + W_BASE + slot * B_BYTES + (j * 256 + w * 64) * 16);
        else
            glds16(fetch ? src : p.zeros, smem + W_BASE + slot * B_BYTES + (j * 256 + w * 64) * 16);
    };

    // ---- consumer: fragment addresses ----
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned c0 = (unsigned)(((0 * 4 + lq) ^ (l15 & 7)) * 16), c1 = (unsigned)(((1 * 4 + lq) ^ (l15 & 7)) * 16);
    const unsigned sm0 = (unsigned)(size_t)smem;
    const unsigned a_off = (unsigned)((wr * MF * 16 + l15) * 128), b_off = (unsigned)(W_BASE + (wc * 160 + l15) * 128);

    f4 acc[MF][10];
    h8 a0[MF], wq[4];   // carried across K-tiles: A fragments of K-step 0 and the first three W fragments of the coming K-tile

    // One K-tile: 20 groups (K-step x W fragment) of MF MFMAs, in the exact order they should issue.  Fillers behind group g:
    //  * W fragment g + 3 (ring of four), A fragments of K-step 1 behind groups 3..8;
    //  * LDS-DMA pieces: the ten W pieces of the NEXT K-tile behind groups 0..4 (two each), then the MF A pieces of the K-tile after it
    //    behind groups 5..10;
    //  * behind group 17: every read of this K-tile has been issued -> lgkmcnt(0), vmcnt(MF) (this wave's W and A pieces of the next
    //    K-tile have landed), s_barrier (everyone's have, and everyone is done reading this K-tile's slots), then the next K-tile's
    //    first fragments, which land under the last 2 MF MFMAs.
    auto ktile = [&](auto first_tag, int sa, int sw, bool fetch_w, bool fetch_a) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const int sa1 = sa == NA - 1 ? 0 : sa + 1;                     // A slot of the next K-tile
        const int sa2 = ALA == 1 ? sa1 : (sa == 0 ? NA - 1 : sa - 1);   // A slot the pieces issued in this body go to (K-tile + ALA)
        const unsigned sba = sm0 + sa * A_BYTES, sbw = sm0 + sw * B_BYTES;
        const unsigned abase[2] = {sba + a_off + c0, sba + a_off + c1};
        const unsigned bbase[2] = {sbw + b_off + c0, sbw + b_off + c1};
        const unsigned nabase = sm0 + sa1 * A_BYTES + a_off + c0, nbbase = sm0 + (sw ^ 1) * B_BYTES + b_off + c0;
        h8 a1[MF];
        int seq = 0, a1_seq = 0, w_seq[23] = {};   // issue order of this K-tile's reads (the carried ones are complete: 0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 20; ++g) {
            const int ks = g / 10, nf = g % 10;
            {
                int need = w_seq[g];
                if (g == 10 && a1_seq > need) need = a1_seq;
                if (need > 0 && g < 18) sw_lgkm(seq - need);   // (groups 18 / 19: covered by the lgkmcnt(0) in front of the barrier)
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const h8& af = ks == 0 ? a0[mf] : a1[mf];
                if constexpr (KO == 3) {
                    if (FIRST && ks == 0) acc[mf][nf] = (f4){0.f, 0.f, 0.f, 0.f};
                    asm volatile("" ::"v"(wq[g & 3]), "v"(af));
                } else if (FIRST && ks == 0)
                    acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[g & 3], af, (f4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                else
                    acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[g & 3], af, acc[mf][nf], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g >= 3 && g < 3 + MF) {
                a1[g - 3] = sw_frag<KO != 2>(abase[1], (g - 3) * 2048);
                a1_seq = ++seq;
            }
            if (g + 3 < 20) {
                const int t = g + 3;
                wq[t & 3] = sw_frag<KO != 2>(bbase[t / 10], (t % 10) * 2048);
                w_seq[t] = ++seq;
            }
            if constexpr (ALA == 2) {   // W first: the A pieces must be the youngest loads at the counted wait
                if (g < 5) {
                    w_piece(2 * g, sw ^ 1, fetch_w);
                    w_piece(2 * g + 1, sw ^ 1, fetch_w);
                } else if (g < 5 + MF) {
                    a_piece(g - 5, sa2, fetch_a);
                }
            } else {                    // A first (the operand that misses gets the longest run-up; measured: W first costs 15-20 %)
                if (2 * g + 1 < MF) {
                    a_piece(2 * g, sa2, fetch_a);
                    a_piece(2 * g + 1, sa2, fetch_a);
                } else if (g == MF / 2) {
                    w_piece(0, sw ^ 1, fetch_w);
                    w_piece(1, sw ^ 1, fetch_w);
                } else if (g > MF / 2 && g - MF / 2 + 1 < 10) {
                    w_piece(g - MF / 2 + 1, sw ^ 1, fetch_w);
                }
            }
            if (g == 13) stream_bump();
            if (g == 17) {
                if constexpr (ALA == 2) {
                    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                    static_assert(MF == 6, "the vmcnt above leaves exactly the MF youngest loads (the A pieces) in flight");
                } else {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) a0[mf] = sw_frag<KO != 2>(nabase, mf * 2048);
                wq[0] = sw_frag<KO != 2>(nbbase, 0);
                wq[1] = sw_frag<KO != 2>(nbbase, 2048);
            }
            if (g == 18) wq[2] = sw_frag<KO != 2>(nbbase, 2 * 2048);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    int tile, kb = 0;   // the segment being multiplied: K-tiles [kb, ke) of `tile` (kb = 0, ke = nk unless SK)
    if constexpr (SK) {
        if (u1 <= u0) return;
        tile = u0 / nk;
        kb = u0 - tile * nk;
    } else {
        tile = next_valid(b0);
        if (tile >= ntiles) return;
    }
    // prologue: A(0) -> A slot 0, W(0) -> W slot 0 (ring form: A(1) -> A slot 1, which may stay in flight)
    a_start(tile, kb);
    w_start(tile, kb);
#pragma unroll
    for (int i = 0; i < MF; ++i) a_piece(i, 0, true);
#pragma unroll
    for (int j = 0; j < 10; ++j) w_piece(j, 0, true);
    stream_bump();
    a_step();
    w_step();
    if constexpr (ALA == 2) {
#pragma unroll
        for (int i = 0; i < MF; ++i) a_piece(i, 1, a_tile < ntiles);
        a_step();
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
        const unsigned na = sm0 + a_off + c0, nb = sm0 + b_off + c0;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) a0[mf] = sw_frag<KO != 2>(na, mf * 2048);
        wq[0] = sw_frag<KO != 2>(nb, 0);
        wq[1] = sw_frag<KO != 2>(nb, 2048);
        wq[2] = sw_frag<KO != 2>(nb, 2 * 2048);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    int sa = 0, sw = 0;   // slots of the K-tile about to be multiplied
    auto body_done = [&]() {
        a_step();
        w_step();
        sa = sa == NA - 1 ? 0 : sa + 1;
        sw ^= 1;
    };

    int u_done = u0;          // SK: first unit of the range not multiplied yet
    bool first_seg = true;    // SK: the segment in hand is the block's first (its partial result goes to slab 2 b, a later one to 2 b + 1)
    while (true) {
        int mt, nt;
        decode(tile, mt, nt);
        const int next_tile = SK ? 0 : next_valid(tile + G);
        const bool has_next = next_tile < ntiles;
        // (K-tile 0 is peeled: its first K-step starts the accumulators from the constant 0, so there is one straight-line definition
        //  of the accumulators per tile and no zeroing pass; dispatch guarantees nk >= 2)
        const int ke = SK ? (nk - kb < u1 - u_done ? nk : kb + (u1 - u_done)) : nk;
        ktile(std::true_type{}, sa, sw, w_tile < ntiles, a_tile < ntiles);
        body_done();
        for (int kt = kb + 1; kt < ke; ++kt) {
            ktile(std::false_type{}, sa, sw, w_tile < ntiles, a_tile < ntiles);
            body_done();
        }

        const int m_wave = mt * BM + wr * MF * 16;
        const int n_wave = nt * BN + wc * 160;
        if (SK && !(kb == 0 && ke == nk)) {
            // ---------------- stream-K: this block holds only K-tiles [kb, ke) of the tile -> raw fp32 accumulators to its slab ----------------
            // slab [192][320] fp32, column = W row inside the N-tile (the order bias / temb / GEGLU pairing are defined in)
            float* slab = p.partial + (size_t)(2 * b0 + (first_seg ? 0 : 1)) * (BM * BN);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                float* srow = slab + (size_t)(wr * MF * 16 + mf * 16 + l15) * BN + wc * 160;
#pragma unroll
                for (int nf = 0; nf < 10; ++nf) {
                    int col;
                    if constexpr (GEGLU)
                        col = nf < 8 ? 64 * (nf >> 2) + 32 * (lq >> 1) + 16 * (nf & 1) + 8 * (lq & 1) + 4 * ((nf >> 1) & 1) : 128 + 16 * (nf - 8) + 4 * lq;
                    else
                        col = 32 * (nf >> 1) + 8 * lq + 4 * (nf & 1);
                    f4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = sw_acc(acc[mf][nf][e]);
                    *(f4*)(srow + col) = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
        // ---------------- epilogue: straight from the accumulators, 16-byte stores (see the W row permutation above) ----------------
        if constexpr (!GEGLU) {
            h8 bias8[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) bias8[q] = *(const h8*)(p.bias != nullptr ? p.bias + n_wave + 32 * q + 8 * lq : p.zeros);
            // residual: a ring of three 16-row slabs, requested two slabs ahead of their use and BEFORE the stores of the slab in
            // between (the wait for a slab then never has to drain younger stores)
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
            auto rows = [&](auto rv_tag) {   // (the temb switch is hoisted: one wave-uniform branch per tile)
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
                            float v0 = sw_acc(acc[mf][2 * q][e]) + (float)bias8[q][e], v1 = sw_acc(acc[mf][2 * q + 1][e]) + (float)bias8[q][4 + e];
                            if constexpr (HAS_RV) {
                                v0 += (float)tv[q][e];
                                v1 += (float)tv[q][4 + e];
                            }
                            o[e] = (half_t)v0;
                            o[4 + e] = (half_t)v1;
                        }
                        if constexpr (EPI == 1) o = o + rr[mf % 3][q];   // fp16 add: correctly rounded, == the fp32 add + rounding of two fp16 values
                        if (ok) *(h8*)(crow + 32 * q) = o;
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one 16-row slab at a time: 40 accumulators out of the AGPRs, not 240
                }
            };
            if constexpr (EPI == 1) {   // (dispatch: no launch carries both a residual and a temb row vector)
                rows(std::false_type{});
            } else {
                if (p.rowvec != nullptr)
                    rows(std::true_type{});
                else
                    rows(std::false_type{});
            }
        } else {
            // fragments 2 np (h) / 2 np + 1 (gate); np 0..3 interleave to 8 consecutive outputs per lane, np 4 is natural order
            const int n_out_wave = n_wave / 2;
            h4 bh[5], bg[5];
#pragma unroll
            for (int np = 0; np < 5; ++np) {
                const int G2 = np >> 1, B = np & 1;
                const int wrow = np < 4 ? 64 * G2 + 32 * (lq >> 1) + 8 * (lq & 1) + 4 * B : 128 + 4 * lq;
                bh[np] = *(const h4*)(p.bias != nullptr ? p.bias + n_wave + wrow : p.zeros);
                bg[np] = *(const h4*)(p.bias != nullptr ? p.bias + n_wave + wrow + 16 : p.zeros);
            }
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int m = m_wave + mf * 16 + l15;
                const bool ok = m < p.M;
                half_t* crow = p.C + (size_t)m * p.ldc + n_out_wave;
                h4 o[5];
#pragma unroll
                for (int np = 0; np < 5; ++np) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float hv = sw_acc(acc[mf][2 * np][e]) + (float)bh[np][e];
                        const float gv = sw_acc(acc[mf][2 * np + 1][e]) + (float)bg[np][e];
                        o[np][e] = (half_t)(hv * av_gelu(gv));
                    }
                }
                if (ok) {
                    *(h8*)(crow + 8 * lq) = __builtin_shufflevector(o[0], o[1], 0, 1, 2, 3, 4, 5, 6, 7);
                    *(h8*)(crow + 32 + 8 * lq) = __builtin_shufflevector(o[2], o[3], 0, 1, 2, 3, 4, 5, 6, 7);
                    *(h4*)(crow + 64 + 4 * lq) = o[4];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (SK) {
            u_done += ke - kb;
            if (u_done >= u1) break;
            ++tile;
            kb = 0;
            first_seg = false;
        } else {
            if (!has_next) break;
            tile = next_tile;
        }
    }
}

// stream-K second pass: one block per output tile; tiles a single range covers were finished by the main kernel (nothing to do),
// the others are the sum of their contributors' slabs in block (= K) order, then the usual epilogue.  `G` = blocks of the main launch.
template <bool GEGLU>
__global__ __launch_bounds__(256) void gemm_sw_fixup_kernel(const GemmK p, int G, int nk) {
    constexpr int BM = SW_BM, BN = SW_BN;
    const int t = blockIdx.x;
    const int tilesM = (p.M + BM - 1) / BM;
    const long long U = (long long)tilesM * p.tilesN * nk;
    auto u_of = [&](int b) { return (int)(U * b / G); };
    auto block_of = [&](int u) {   // the b with u_of(b) <= u < u_of(b + 1)
        int b = (int)(((long long)(u + 1) * G - 1) / U);
        while (b > 0 && u_of(b) > u) --b;
        while (b + 1 < G && u_of(b + 1) <= u) ++b;
        return b;
    };
    const int bf = block_of(t * nk), bl = block_of((t + 1) * nk - 1);
    if (bf == bl) return;
    const int mt = t / p.tilesN, nt = t - mt * p.tilesN;
    const int NO = GEGLU ? BN / 2 : BN;          // output columns of the tile
    const int Nout = GEGLU ? p.N / 2 : p.N;
    for (int id = threadIdx.x; id < BM * (NO / 8); id += 256) {
        const int r = id / (NO / 8), c8 = (id - r * (NO / 8)) * 8;
        const int m = mt * BM + r;
        if (m >= p.M) continue;
        // W-row columns of the 8 outputs: plain = c8 .. c8 + 7; GEGLU: output j <- h column 32 (j / 16) + j % 16, gate column + 16
        const int ch = GEGLU ? 32 * (c8 >> 4) + (c8 & 15) : c8;
        float vh[8], vg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vh[e] = vg[e] = 0.f;
        for (int b = bf; b <= bl; ++b) {
            const int first_tile = u_of(b) / nk;
            const float* slab = p.partial + (size_t)(2 * b + (t == first_tile ? 0 : 1)) * (BM * BN) + (size_t)r * BN;
            const f4 a0 = *(const f4*)(slab + ch), a1 = *(const f4*)(slab + ch + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vh[e] += a0[e];
                vh[4 + e] += a1[e];
            }
            if constexpr (GEGLU) {
                const f4 g0 = *(const f4*)(slab + ch + 16), g1 = *(const f4*)(slab + ch + 20);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vg[e] += g0[e];
                    vg[4 + e] += g1[e];
                }
            }
        }
        const int n_w = nt * BN + ch;            // W row / bias index of the first value
        h8 o;
        if constexpr (GEGLU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float hv = vh[e] + (p.bias != nullptr ? (float)p.bias[n_w + e] : 0.f);
                const float gv = vg[e] + (p.bias != nullptr ? (float)p.bias[n_w + 16 + e] : 0.f);
                o[e] = (half_t)(hv * av_gelu(gv));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = vh[e] + (p.bias != nullptr ? (float)p.bias[n_w + e] : 0.f);
                if (p.rowvec != nullptr) v += (float)p.rowvec[(size_t)(m / p.rowvec_div) * p.ldrv + n_w + e];
                o[e] = (half_t)v;
            }
            if (p.R != nullptr) o = o + *(const h8*)(p.R + (size_t)m * p.ldr + n_w);
        }
        const int n_out = (GEGLU ? nt * (BN / 2) : nt * BN) + c8;
        if (n_out < Nout) *(h8*)(p.C + (size_t)m * p.ldc + n_out) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------
bool av_gemm_sw_eligible(const AnyV2VGemmDesc* d) {
    const bool geglu = d->act == ACT_GEGLU;
    const int nk = (d->mode == MODE_LINEAR ? 1 : (d->mode == MODE_CONV2D ? 9 : 3)) * ((d->C0 + d->C1) / 64);
    return d->N % 320 == 0 && (geglu ? d->mode == MODE_LINEAR && d->R == nullptr && d->rowvec == nullptr : d->act == ACT_NONE) &&
           !(d->R != nullptr && d->rowvec != nullptr) && nk >= 2 && d->ldc % 8 == 0 && (d->R == nullptr || d->ldr % 8 == 0) &&
           (d->rowvec == nullptr || d->ldrv % 8 == 0);
}

template <int MODE>
static void sw_launch_mode(const GemmK& k, const AnyV2VGemmDesc* d, dim3 grid, hipStream_t s) {
    if constexpr (MODE == MODE_LINEAR) {
        if (d->act == ACT_GEGLU) {
            hipLaunchKernelGGL((gemm_sw_kernel<MODE_LINEAR, 2>), grid, dim3(256), 0, s, k);
            return;
        }
    }
#ifdef ANYV2V_EXPERIMENTS
    if constexpr (MODE == MODE_CONV2D) {
        const int ko = (d->flags >> 23) & 15;
        if (ko == 7 && d->R == nullptr) { hipLaunchKernelGGL((gemm_sw_kernel<MODE, 0, 7>), grid, dim3(256), 0, s, k); return; }
        if (ko == 8 && d->R == nullptr) { hipLaunchKernelGGL((gemm_sw_kernel<MODE, 0, 8>), grid, dim3(256), 0, s, k); return; }
    }
    if constexpr (MODE == MODE_LINEAR) {
        const int ko = (d->flags >> 23) & 7;
        if (ko == 1) { hipLaunchKernelGGL((gemm_sw_kernel<MODE, 0, 1>), grid, dim3(256), 0, s, k); return; }
        if (ko == 2) { hipLaunchKernelGGL((gemm_sw_kernel<MODE, 0, 2>), grid, dim3(256), 0, s, k); return; }
        if (ko == 3) { hipLaunchKernelGGL((gemm_sw_kernel<MODE, 0, 3>), grid, dim3(256), 0, s, k); return; }
        if (ko == 4) { hipLaunchKernelGGL((gemm_sw_kernel<MODE, 0, 4>), grid, dim3(256), 0, s, k); return; }
        if (ko == 5) { hipLaunchKernelGGL((gemm_sw_kernel<MODE, 0, 5>), grid, dim3(256), 0, s, k); return; }
        if (ko == 6) { hipLaunchKernelGGL((gemm_sw_kernel<MODE, 0, 6>), grid, dim3(256), 0, s, k); return; }
    }
#endif
    if (d->R != nullptr)
        hipLaunchKernelGGL((gemm_sw_kernel<MODE, 1>), grid, dim3(256), 0, s, k);
    else
        hipLaunchKernelGGL((gemm_sw_kernel<MODE, 0>), grid, dim3(256), 0, s, k);
}

// k: filled by anyv2v_gemm_f16 (tilesN / raster fields set here)
int av_gemm_sw_launch(GemmK& k, const AnyV2VGemmDesc* d, hipStream_t s) {
    const int tiles = ((d->M + SW_BM - 1) / SW_BM) * (d->N / 320);
    k.tilesN = d->N / 320;
    int gmax = 256;
#ifdef ANYV2V_EXPERIMENTS   // probe (tools/gemm_sw_grid_probe.py): fewer persistent blocks = fewer CUs share the L2 / fabric
    if (const char* e = getenv("ANYV2V_SW_GRID")) gmax = atoi(e) > 0 ? atoi(e) : 256;
#endif
    const dim3 grid(tiles < gmax ? tiles : gmax);
    {   // tile order of wide-N launches: as gemm_big_kernel's dispatch (flags bits 13-16)
        const int code = (d->flags >> 13) & 7;
        static const int gm_of[8] = {0, 0, 4, 8, 16, 32, 2, 0};
        int gm = gm_of[code];
        if (code == 0 && k.tilesN >= 8 && tiles >= 512) gm = 8;
        const int tm = (d->M + SW_BM - 1) / SW_BM;
        if (gm > 0 && grid.x == 256 && k.tilesN % (32 / gm) == 0) {
            k.rast_gm = gm;
            k.rast_gn = 32 / gm;
            k.rast_sm = (tm + gm - 1) / gm;
            k.rast_sn = k.tilesN / k.rast_gn;
            k.rast_nfast = (d->flags >> 16) & 1;
        }
    }
    if (d->mode == MODE_CONV2D)
        sw_launch_mode<MODE_CONV2D>(k, d, grid, s);
    else if (d->mode == MODE_TEMPORAL)
        sw_launch_mode<MODE_TEMPORAL>(k, d, grid, s);
    else
        sw_launch_mode<MODE_LINEAR>(k, d, grid, s);
    return av_launch_status("gemm_sw");
}

// ---- stream-K form -------------------------------------------------------------------------------------
// Workspace: two fp32 slabs of 192 x 320 per block of the main launch.
size_t av_gemm_sw_sk_workspace(int blocks) { return (size_t)2 * blocks * SW_BM * SW_BN * sizeof(float); }

// Blocks the stream-K form would use, or 0 when it should not be taken: the launch's (tile, K-tile) units are dealt evenly to
// min(256, units / 4) blocks.  It pays where whole tiles quantise badly onto 256 CUs (tiles / (rounds x 256) below ~0.9) and every
// block still gets a few K-tiles.
int av_gemm_sw_sk_blocks(const AnyV2VGemmDesc* d, bool force) {
    const int tiles = ((d->M + SW_BM - 1) / SW_BM) * (d->N / 320);
    const int nk = (d->mode == MODE_LINEAR ? 1 : (d->mode == MODE_CONV2D ? 9 : 3)) * ((d->C0 + d->C1) / 64);
    const long long U = (long long)tiles * nk;
    if (U < 256 * 4 || U > (1ll << 22)) return force && U >= 8 ? (int)(U / 4 < 256 ? U / 4 : 256) : 0;
    if (force) return 256;
    const int rounds = (tiles + 255) / 256;
    const double eff = (double)tiles / (rounds * 256.0);
    return eff < 0.9 ? 256 : 0;
}

template <int MODE>
static void sw_sk_launch_mode(const GemmK& k, const AnyV2VGemmDesc* d, dim3 grid, hipStream_t s) {
    if constexpr (MODE == MODE_LINEAR) {
        if (d->act == ACT_GEGLU) {
            hipLaunchKernelGGL((gemm_sw_kernel<MODE_LINEAR, 2, 0, true>), grid, dim3(256), 0, s, k);
            return;
        }
    }
    if (d->R != nullptr)
        hipLaunchKernelGGL((gemm_sw_kernel<MODE, 1, 0, true>), grid, dim3(256), 0, s, k);
    else
        hipLaunchKernelGGL((gemm_sw_kernel<MODE, 0, 0, true>), grid, dim3(256), 0, s, k);
}

int av_gemm_sw_sk_launch(GemmK& k, const AnyV2VGemmDesc* d, int blocks, hipStream_t s) {
    const int tiles = ((d->M + SW_BM - 1) / SW_BM) * (d->N / 320);
    const int nk = k.taps * (k.nt0 + k.nt1);
    k.tilesN = d->N / 320;
    k.partial = (float*)d->workspace;
    const dim3 grid(blocks);
    if (d->mode == MODE_CONV2D)
        sw_sk_launch_mode<MODE_CONV2D>(k, d, grid, s);
    else if (d->mode == MODE_TEMPORAL)
        sw_sk_launch_mode<MODE_TEMPORAL>(k, d, grid, s);
    else
        sw_sk_launch_mode<MODE_LINEAR>(k, d, grid, s);
    if (d->act == ACT_GEGLU)
        hipLaunchKernelGGL((gemm_sw_fixup_kernel<true>), dim3(tiles), dim3(256), 0, s, k, blocks, nk);
    else
        hipLaunchKernelGGL((gemm_sw_fixup_kernel<false>), dim3(tiles), dim3(256), 0, s, k, blocks, nk);
    return av_launch_status("gemm_sw<stream-K>");
}
