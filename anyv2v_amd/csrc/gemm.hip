// Gather-GEMM for gfx950: Linear / 1x1 conv, implicit-GEMM conv2d 3x3 (stride 1|2, optional folded nearest x2
// upsample) and the temporal (3,1,1) conv, all over channels-last token matrices, fp16 in / fp32 MFMA
// accumulate / fp16 out, with the bias / temb-broadcast / SiLU / GELU / GEGLU / residual epilogues fused.
//
// Replaces (reference = TIGER-AI-Lab/AnyV2V, i2vgen-xl/pnp_utils.py): conv1/conv2 :78,:107, time_emb_proj :81-88,
// conv_shortcut :117-122, residual :124, attn.to_q/to_k/to_v :175,:182-183, attn.to_out[0] :216, and the
// diffusers-0.26.3 Linear/Conv2d/Conv3d layers of I2VGenXLUNet behind pipeline_i2vgen_xl.py:1146.
//
// Wave tile: 64 x NF*16 via v_mfma_f32_16x16x32_f16 with SWAPPED operands (a = weight fragment, b = activation
// fragment) so that a lane ends up with 4 consecutive output channels of one token -> 8-byte LDS writes in the
// epilogue and full-line coalesced 16-byte global stores.  LDS tiles are [row][64 k] with the 16-byte chunk index
// XOR-swizzled by (row & 7): conflict-free for the ds_read_b128 fragment reads (MI355X guide, T2); with LDS-DMA the
// swizzle is applied on the global SOURCE address (destination stays lane-linear, guide rule 21).
//   gemm_mfma_kernel  : 128 x NF*32 x 64, 4 waves 2x2, 2 LDS stages (register- or LDS-DMA-staged), 2 blocks/CU.
//   gemm_big_kernel   : 256 x 320 x 64, 8 waves 4x2 (wave tile 64 x 160), 2 LDS stages by LDS-DMA, persistent
//                       blocks with cross-tile prefetch and a wave-private epilogue -- the large-M workhorse.
#include <stdlib.h>
#include <type_traits>

#include "gemm_common.h"

__device__ __attribute__((aligned(256))) half_t g_zero_line[128];  // 256 B of zeros: source for padded taps

// ---------------------------------------------------------------------------------------------------------
// Shared epilogue: accumulators -> (+bias, +temb row vector, activation / GEGLU) -> fp16 tile staged in LDS ->
// (+residual) -> coalesced 16-byte stores.  Caller guarantees all waves are done with the pipeline LDS.
template <int NF, bool GEGLU, int BM, int NTHREADS>
__device__ __forceinline__ void epilogue(const GemmK& p, f4 (&acc)[4][NF], char* smem, int m_blk, int n_blk, int wr,
                                         int wc, int lane, int tid, int split = 0, long long* tr = nullptr) {
    constexpr int BN = NF * 32;
    constexpr int BNO = GEGLU ? BN / 2 : BN;
    constexpr int CS_LD = BNO + 8;
    half_t* const Cs = (half_t*)smem;
    const int l15 = lane & 15, lq = lane >> 4;
    if (p.splits > 1) {  // split-K: raw fp32 partial tile; bias / temb / activation / residual happen in the reduce kernel
        float* dst = p.partial + (size_t)split * p.M * p.N;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int m = m_blk + wr * 64 + mf * 16 + l15;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int n = n_blk + wc * NF * 16 + nf * 16 + 4 * lq;
                if (m < p.M && n + 4 <= p.N) *(f4*)(dst + (size_t)m * p.N + n) = acc[mf][nf];
            }
        }
        return;
    }
    if (p.act == ACT_F32OUT) {  // raw fp32 result (+bias): attention logits of the VAE's 512-wide single head
        float* dst = (float*)p.C;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int m = m_blk + wr * 64 + mf * 16 + l15;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int n = n_blk + wc * NF * 16 + nf * 16 + 4 * lq;
                if (m < p.M && n + 4 <= p.N) {
                    f4 v = acc[mf][nf];
                    if (p.bias != nullptr) {
                        const h4 b = *(const h4*)(p.bias + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)b[r];
                    }
                    *(f4*)(dst + (size_t)m * p.ldc + n) = v;
                }
            }
        }
        return;
    }
    const int Nout = GEGLU ? p.N / 2 : p.N;
    const int n_out_blk = GEGLU ? n_blk / 2 : n_blk;
    constexpr int CPR = BNO / 8;                // 16-byte chunks per output-tile row
    constexpr int NIT = BM * CPR / NTHREADS;    // chunks per thread in the store phase
    static_assert(BM * CPR % NTHREADS == 0, "store phase assumes an exact chunk split");
    const bool full_chunks = (Nout & 7) == 0;   // wave-uniform; false only for the tiny-N layers (conv_out, N = 4)
    // Every global operand of the epilogue is requested up front, in one batch, so that their latencies overlap each
    // other and the convert / LDS-staging work below (in-kernel timestamps showed the previous form -- loads next to
    // their consumers -- spending 5-7 us per block in serialized L2 round trips, and 8-9 us in the residual loop):
    //   residual chunks of the store phase -> rr[], bias -> bvec[], temb row vector -> tvec[][] (only when present).
    h8 rr[NIT];
    if (p.R != nullptr && full_chunks) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int id = tid + it * NTHREADS;
            const int r = id / CPR, cc = id - r * CPR;
            const int m = m_blk + r, n0 = n_out_blk + cc * 8;
            rr[it] = *(const h8*)((m < p.M && n0 < Nout) ? p.R + (size_t)m * p.ldr + n0 : p.zeros);
        }
    }
    h4 bvec[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const int n = n_blk + wc * NF * 16 + nf * 16 + 4 * lq;
        const half_t* src = (p.bias != nullptr && n + 4 <= p.N) ? p.bias + n : p.zeros;
        bvec[nf] = *(const h4*)src;
    }
    h4 tvec[GEGLU ? 1 : 4][GEGLU ? 1 : NF];
    const bool has_rowvec = !GEGLU && p.rowvec != nullptr;
    if constexpr (!GEGLU) {
        if (has_rowvec) {
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                const int m = m_blk + wr * 64 + mf * 16 + l15;
                const bool ok = m < p.M;
                const half_t* rv = p.rowvec + (size_t)((ok ? m : 0) / p.rowvec_div) * p.ldrv;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    const int n = n_blk + wc * NF * 16 + nf * 16 + 4 * lq;
                    tvec[mf][nf] = *(const h4*)((ok && n + 4 <= p.N) ? rv + n : p.zeros);
                }
            }
        }
    }
    if constexpr (GEGLU) {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int ml = wr * 64 + mf * 16 + l15;
#pragma unroll
            for (int np = 0; np < NF / 2; ++np) {
                h4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // h * gelu(gate) in fp32, ONE rounding (torch's fp16 path rounds proj, gelu and the product; its fp32 path --
                    // the reference this is checked against -- none of them)
                    const float hv = acc[mf][2 * np][r] + (float)bvec[2 * np][r];
                    const float gv = acc[mf][2 * np + 1][r] + (float)bvec[2 * np + 1][r];
                    o[r] = (half_t)(hv * av_gelu(gv));
                }
                *(h4*)(Cs + ml * CS_LD + wc * NF * 8 + np * 16 + 4 * lq) = o;
            }
        }
    } else {
        // The activation switch is hoisted out of the element loops (one wave-uniform branch per tile): left inside,
        // hipcc if-converts it and evaluates SiLU *and* erf-GELU for all 80 outputs of every thread (measured 5-7 us
        // per block on plain linear layers).
        auto stage = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                const int ml = wr * 64 + mf * 16 + l15;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    const int nl = wc * NF * 16 + nf * 16 + 4 * lq;
                    h4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[mf][nf][r] + (float)bvec[nf][r];
                        if (has_rowvec) v += (float)tvec[mf][nf][r];
                        if constexpr (ACT == ACT_SILU) v = av_silu(v);
                        if constexpr (ACT == ACT_GELU) v = av_gelu(v);
                        o[r] = (half_t)v;
                    }
                    *(h4*)(Cs + ml * CS_LD + nl) = o;
                }
            }
        };
        if (p.act == ACT_SILU)
            stage(std::integral_constant<int, ACT_SILU>{});
        else if (p.act == ACT_GELU)
            stage(std::integral_constant<int, ACT_GELU>{});
        else
            stage(std::integral_constant<int, ACT_NONE>{});
    }
    if (tr != nullptr && tid == 0) tr[24] = (long long)__builtin_amdgcn_s_memtime();
    __syncthreads();
    if (tr != nullptr && tid == 0) tr[25] = (long long)__builtin_amdgcn_s_memtime();
    if (full_chunks) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int id = tid + it * NTHREADS;
            const int r = id / CPR, cc = id - r * CPR;
            const int m = m_blk + r, n0 = n_out_blk + cc * 8;
            h8 v = *(const h8*)(Cs + r * CS_LD + cc * 8);
            if (p.R != nullptr) {
                v = v + rr[it];  // fp16 add: correctly rounded, i.e. what the fp32 add + rounding of two fp16 values gives
            }
            if (m < p.M && n0 < Nout) *(h8*)(p.C + (size_t)m * p.ldc + n0) = v;
        }
        return;
    }
    for (int id = tid; id < BM * CPR; id += NTHREADS) {  // ragged N: element-wise tail
        const int r = id / CPR, cc = id - r * CPR;
        const int m = m_blk + r;
        const int n0 = n_out_blk + cc * 8;
        if (m >= p.M || n0 >= Nout) continue;
        const h8 v = *(const h8*)(Cs + r * CS_LD + cc * 8);
        for (int e = 0; e < 8 && n0 + e < Nout; ++e) {
            float x = (float)v[e];
            if (p.R != nullptr) x += (float)p.R[(size_t)m * p.ldr + n0 + e];
            p.C[(size_t)m * p.ldc + n0 + e] = (half_t)x;
        }
    }
}

// Incremental gather addressing: inside one (tap, source) run consecutive K-tiles only advance the channel offset
// (+128 bytes); the row -> shifted-row math is redone only when the tap or the source changes (wave-uniform branch).
template <int MODE>
struct AGen {
    const half_t* ap[4];
    int astep[4];  // halves to advance per K-tile: 64, or 0 for rows that read the zero line
    int ktc, tap;
    __device__ __forceinline__ void recompute(const GemmK& p, const RowInfo (&ri)[4], int kc) {
        const ASrc s = a_source(p, ktc, kc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sr = src_row<MODE>(p, ri[i], tap);
            ap[i] = a_addr(p, s, sr);
            astep[i] = sr < 0 ? 0 : 64;
        }
    }
    __device__ __forceinline__ void start(const GemmK& p, const RowInfo (&ri)[4], int kc, int kt0 = 0, int ntap = 1) {
        tap = kt0 / ntap;
        ktc = kt0 - tap * ntap;
        recompute(p, ri, kc);
    }
    __device__ __forceinline__ void next(const GemmK& p, const RowInfo (&ri)[4], int kc, int ntap) {
        if (++ktc == ntap) {
            ktc = 0;
            ++tap;
        }
        if (ktc == 0 || ktc == p.nt0) {
            recompute(p, ri, kc);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) ap[i] += astep[i];
        }
    }
};

#ifdef ANYV2V_EXPERIMENTS
#include "../../tools/experiments/gemm_probe_config.h"   // AV_TRACE_TILE (which tile of a block the probe stamps)
#else
constexpr int AV_TRACE_TILE = 0;
#endif
// The fragment reads below are inline asm with hand-counted waits; what that relies on is checked over the generated assembly by
// tests/test_isa_guards.py (no scratch access / copy of a pending destination, every MFMA covered by its counted wait).  Validated
// with ROCm 7.2's hipcc only: a different compiler may schedule around the asm differently -- rerun that test and `-m gpu`.
#if defined(HIP_VERSION_MAJOR) && (HIP_VERSION_MAJOR != 7 || HIP_VERSION_MINOR != 2)
#warning "gemm.hip: inline-asm LDS fragment reads were validated with ROCm 7.2 only; rerun tests/test_isa_guards.py and the -m gpu suite"
#endif
// Fragment reads of the K-tile below are issued as inline asm with hand-counted `s_waitcnt lgkmcnt(n)`: with an LDS-DMA load
// (global_load_lds) in flight hipcc treats the LGKM counter as out of order and waits lgkmcnt(0) before every fragment use,
// i.e. also for the fragment it has just requested two groups ahead -- the roll degenerates into issue -> full LDS latency ->
// use (tools/wait_probe.hip reproduces it in 30 lines).  LDS reads return in order among themselves, and the DMA completes on
// vmcnt, so the wait a use needs is "all but the reads issued after mine".
__device__ __forceinline__ h8 lds_frag(unsigned base, int off) {  // off: a constant after unrolling (16-bit immediate)
    h8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(off) : "memory");
    return v;
}
__device__ __forceinline__ void lgkm_wait(int n) {  // n is a constant after unrolling; the switch folds to one s_waitcnt
    switch (n) {
#define AV_LGW(k) case k: asm volatile("s_waitcnt lgkmcnt(" #k ")" ::: "memory"); break;
        AV_LGW(0) AV_LGW(1) AV_LGW(2) AV_LGW(3) AV_LGW(4) AV_LGW(5) AV_LGW(6) AV_LGW(7) AV_LGW(8) AV_LGW(9) AV_LGW(10)
        AV_LGW(11) AV_LGW(12) AV_LGW(13) AV_LGW(14) AV_LGW(15)
#undef AV_LGW
        default: asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); break;  // (more than 15 younger reads: the counter saturates there)
    }
}

// one K-tile (64) of MFMA work for a 64 x NF*16 wave tile: all 2*(4+NF) fragment reads are issued first, so the
// compiler can retire them with counted lgkmcnt waits while the MFMAs of the first K-step already run (loading per
// K-step made it emit a full lgkmcnt(0) in front of every MFMA batch).
// KO (debug knock-outs, tools/gemm_trace.py): 4 = no fragment reads (register constants), 5 = no MFMAs
template <int NF, int KO = 0>
__device__ __forceinline__ void mma_tile(f4 (&acc)[4][NF], const char* as, const char* bs, int wr, int wc, int lane) {
    const int l15 = lane & 15, lq = lane >> 4;
    h8 af[2][4], bf[2][NF];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = (ks * 4 + lq) ^ (l15 & 7);
        if constexpr (KO == 4) {
            h8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (half_t)(float)(lane + e);
            asm volatile("" : "+v"(x));
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) bf[ks][nf] = x;
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) af[ks][mf] = x;
        } else {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) bf[ks][nf] = *(const h8*)(bs + ((wc * NF * 16 + nf * 16 + l15) * 8 + c) * 16);
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) af[ks][mf] = *(const h8*)(as + ((wr * 64 + mf * 16 + l15) * 8 + c) * 16);
        }
    }
    if constexpr (KO == 5) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) asm volatile("" ::"v"(bf[ks][nf]));
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) asm volatile("" ::"v"(af[ks][mf]));
        }
        return;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][nf], af[ks][mf], acc[mf][nf], 0, 0, 0);
    if constexpr (KO == 4) {
        __builtin_amdgcn_sched_barrier(0);
        return;
    }
    // scheduling contract for this region: all fragment reads first, then the MFMAs (hipcc otherwise sinks each read
    // next to its consumer and drains with lgkmcnt(0) four to six times per tile)
    // K-step 0 fragments, then K-step 0 MFMAs with the K-step 1 reads slotted in (1 read per 2 MFMAs), then the rest
    __builtin_amdgcn_sched_group_barrier(0x100, 4 + NF, 0);
#pragma unroll
    for (int i = 0; i < 4 + NF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8 * NF - 2 * (4 + NF), 0);
    // keep the MFMAs above the caller's end-of-tile s_waitcnt (an asm "memory" clobber does not order register-only MFMAs)
    __builtin_amdgcn_sched_barrier(0);
}

// The same K-tile with the fragment reads as inline asm and counted waits (LDS-DMA kernel: the next tile's pieces are already in
// flight here, so hipcc would wait lgkmcnt(0) in front of both MFMA batches, see lds_frag): K-step 0's reads, then its MFMAs
// each waiting only for its own two fragments, K-step 1's reads slotted in one per two MFMAs.
template <int NF>
__device__ __forceinline__ void mma_tile_asm(f4 (&acc)[4][NF], const char* as, const char* bs, int wr, int wc, int lane) {
    const int l15 = lane & 15, lq = lane >> 4;
    const int c0 = ((0 * 4 + lq) ^ (l15 & 7)) * 16, c1 = ((1 * 4 + lq) ^ (l15 & 7)) * 16;
    const char* a0 = as + (wr * 64 + l15) * 128;
    const char* b0 = bs + (wc * NF * 16 + l15) * 128;
    const unsigned abase[2] = {(unsigned)(size_t)(a0 + c0), (unsigned)(size_t)(a0 + c1)};
    const unsigned bbase[2] = {(unsigned)(size_t)(b0 + c0), (unsigned)(size_t)(b0 + c1)};
    h8 af[2][4], bf[2][NF];
    int seq = 0, done = 0, a_seq[2][4] = {}, b_seq[2][NF] = {};
#define AV_RA(ks, mf) (af[ks][mf] = lds_frag(abase[ks], (mf) * 2048), a_seq[ks][mf] = ++seq)
#define AV_RB(ks, nf) (bf[ks][nf] = lds_frag(bbase[ks], (nf) * 2048), b_seq[ks][nf] = ++seq)
    AV_RA(0, 0);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) AV_RB(0, nf);
#pragma unroll
    for (int mf = 1; mf < 4; ++mf) AV_RA(0, mf);
    __builtin_amdgcn_sched_barrier(0);
    int slot = 0;  // K-step 1 reads issued so far, in the order a(1,0), b(1,0..NF-1), a(1,1..3)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int need = a_seq[ks][mf] > b_seq[ks][nf] ? a_seq[ks][mf] : b_seq[ks][nf];
                if (need > done) {
                    lgkm_wait(seq - need);
                    done = need;
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][nf], af[ks][mf], acc[mf][nf], 0, 0, 0);
                if (ks == 0 && ((mf * NF + nf) & 1) && slot < 4 + NF) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (slot == 0)
                        AV_RA(1, 0);
                    else if (slot <= NF)
                        AV_RB(1, slot - 1);
                    else
                        AV_RA(1, slot - NF);
                    ++slot;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#undef AV_RA
#undef AV_RB
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------------------
// KO (debug knock-outs): 2 = K loop issues only the W tiles, 3 = K loop issues no loads, 4 / 5 see mma_tile
template <int NF, bool GLDS, bool GEGLU, int MODE, bool TRACE = false, int KO = 0>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(const GemmK p) {
    constexpr int BM = 128, BN = NF * 32;
    constexpr int A_BYTES = BM * 64 * 2;
    constexpr int B_BYTES = BN * 64 * 2;
    constexpr int NB = BN / 32;
    __shared__ __attribute__((aligned(16))) char smem[2 * (A_BYTES + B_BYTES)];
    char* const As0 = smem;
    char* const Bs0 = smem + 2 * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    long long* tr = nullptr;
    if constexpr (TRACE) {
        tr = p.trace + (size_t)blockIdx.x * 32;
        if (tid == 0) {
            tr[0] = ((long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
            tr[1] = (long long)__builtin_amdgcn_s_memrealtime();
            tr[2] = (long long)__builtin_amdgcn_s_memtime();
        }
    }
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);  // XCD-contiguous tile order (bijective)
    const int ntiles = nwg / p.splits;
    const int split = bid / ntiles;  // split-K: this block covers K-tiles [kt_begin, kt_end) of its output tile
    bid -= split * ntiles;
    const int mt = bid / p.tilesN, nt = bid - mt * p.tilesN;
    const int m_blk = mt * BM, n_blk = nt * BN;

    // staging: thread -> rows srow0 + 32 i, physical 16-B chunk pc, logical chunk kc
    const int srow0 = tid >> 3;
    const int pc = tid & 7;
    const int kc = pc ^ (srow0 & 7);
    RowInfo ri[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ri[i] = make_row<MODE>(p, m_blk + srow0 + 32 * i);
    const half_t* bptr[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int n = n_blk + srow0 + 32 * i;
        n = n < p.N ? n : p.N - 1;
        bptr[i] = p.W + (size_t)n * p.Ktot + kc * 8;
    }

    h8 ra[4], rb[NB];
    const int ntap = p.nt0 + p.nt1;
    const int nk_all = p.taps * ntap;
    const int kt_begin = (int)(((long long)nk_all * split) / p.splits);
    const int kt_end = (int)(((long long)nk_all * (split + 1)) / p.splits);
#pragma unroll
    for (int i = 0; i < NB; ++i) bptr[i] += (size_t)kt_begin * 64;
    AGen<MODE> gen;
    gen.start(p, ri, kc, kt_begin, ntap);
    auto issue = [&](int buf, bool with_a = true) {  // loads the generator's current tile, then advances it
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (GLDS) {
                if (with_a) glds16(gen.ap[i], As0 + buf * A_BYTES + (i * 256 + w * 64) * 16);
            } else {
                ra[i] = *(const h8*)gen.ap[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if constexpr (GLDS)
                glds16(bptr[i], Bs0 + buf * B_BYTES + (i * 256 + w * 64) * 16);
            else
                rb[i] = *(const h8*)bptr[i];
            bptr[i] += 64;
        }
        gen.next(p, ri, kc, ntap);
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(h8*)(As0 + buf * A_BYTES + ((srow0 + 32 * i) * 8 + pc) * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *(h8*)(Bs0 + buf * B_BYTES + ((srow0 + 32 * i) * 8 + pc) * 16) = rb[i];
    };

    f4 acc[4][NF];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const int nk = kt_end - kt_begin;
    issue(0);
    if constexpr (!GLDS) commit(0);
    if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (TRACE) if (tid == 0) tr[3] = (long long)__builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool has_next = kt + 1 < nk;
        if (has_next && KO != 3) issue(cur ^ 1, KO != 2);
        if constexpr (GLDS && KO == 0)
            mma_tile_asm<NF>(acc, As0 + cur * A_BYTES, Bs0 + cur * B_BYTES, wr, wc, lane);
        else
            mma_tile<NF, KO>(acc, As0 + cur * A_BYTES, Bs0 + cur * B_BYTES, wr, wc, lane);
        if constexpr (TRACE) if (tid == 0 && kt < 8) tr[4 + kt] = (long long)__builtin_amdgcn_s_memtime();
        if constexpr (!GLDS) {
            if (has_next) commit(cur ^ 1);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if constexpr (TRACE) if (tid == 0 && kt < 8) tr[12 + kt] = (long long)__builtin_amdgcn_s_memtime();
    }
    if constexpr (TRACE) if (tid == 0) tr[20] = (long long)__builtin_amdgcn_s_memtime();
    epilogue<NF, GEGLU, BM, 256>(p, acc, smem, m_blk, n_blk, wr, wc, lane, tid, split, tr);
    if constexpr (TRACE) {
        if (tid == 0) tr[26] = (long long)__builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            tr[21] = (long long)__builtin_amdgcn_s_memtime();
            tr[22] = (long long)__builtin_amdgcn_s_memrealtime();
            tr[23] = nk;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Large-M persistent kernel: 256 x 320 x 64 block tile, 8 waves 4(M) x 2(N), wave tile 64 x 160 (4 x 10 MFMA 16x16x32
// fragments, 160 accumulator registers), two LDS stages of 72 KB filled by LDS-DMA.
//
// Why this shape (measured on the 128-row kernel with in-kernel timestamps and knock-outs, tools/gemm_trace.py):
// removing the MFMAs from its K loop saved 19 %, removing the LDS-DMA loads 45 % -- the loop is bound by the operand
// stream (14 KB of L2->LDS traffic and 14 DMA instructions per MFLOP), not by the matrix cores.  A 256 x 320 tile
// halves both (7 KB and 7 DMA instructions per MFLOP) and cuts fragment re-reads from LDS by 28 %.  All channel
// counts of the UNet are multiples of 320, so the 320-wide tile has no N waste.
//
// One block per CU (147 KB of LDS), grid = min(tiles, 256) persistent blocks walking tiles in XCD-contiguous order.
// The first K-tile of a block's NEXT output tile is requested during the last K-tile of the current one, so the
// prologue latency is paid once per block, and the epilogue runs wave-privately (16-row slabs staged through the
// just-consumed LDS stage, no block barriers) while that prefetch is in flight.
// One K-tile (64) for the 64 x 160 wave tile, written in the exact order it should issue (sched_barrier pins it):
//  * weight fragments roll: bf[nf] is read two fragments ahead of its four MFMAs, so at most three are live; the
//    activation fragments of the next K-step are read during the last four fragment groups (44 fragment registers
//    live next to the 160 accumulators, instead of 112 when hipcc hoists all 28 reads of the tile to the top);
//  * the next tile's LDS-DMA pieces are threaded through the first half of the MFMA stream, one per fragment group
//    (they have to sit here textually: an LDS-DMA load writes LDS, so hipcc never moves it across a ds_read).
template <int MF, typename PieceFn>
__device__ __forceinline__ void mma_tile_big(f4 (&acc)[MF][10], const char* as, const char* bs, int wr, int wc, int lane,
                                             PieceFn&& piece) {
    const int l15 = lane & 15, lq = lane >> 4;
    const char* a0 = as + (wr * MF * 16 + l15) * 128;
    const char* b0 = bs + (wc * 160 + l15) * 128;
    const int c0 = ((0 * 4 + lq) ^ (l15 & 7)) * 16, c1 = ((1 * 4 + lq) ^ (l15 & 7)) * 16;
    h8 af[2][MF], bf[2][10];
    // issue order of the reads (seq = running count) and, per fragment, its position in that order
    int seq = 0, a_seq[2] = {0, 0}, b_seq[2][10] = {};
    const unsigned abase[2] = {(unsigned)(size_t)(a0 + c0), (unsigned)(size_t)(a0 + c1)};
    const unsigned bbase[2] = {(unsigned)(size_t)(b0 + c0), (unsigned)(size_t)(b0 + c1)};
#define AV_RA(ks, mf) (af[ks][mf] = lds_frag(abase[ks], (mf) * 2048), a_seq[ks] = ++seq)
#define AV_RB(ks, nf) (bf[ks][nf] = lds_frag(bbase[ks], (nf) * 2048), b_seq[ks][nf] = ++seq)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) AV_RA(0, mf);
    AV_RB(0, 0);
    AV_RB(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    int npiece = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int nf = 0; nf < 10; ++nf) {
            {   // everything up to the later of (this group's weight fragment, this K-step's last activation fragment)
                const int need = b_seq[ks][nf] > a_seq[ks] ? b_seq[ks][nf] : a_seq[ks];
                lgkm_wait(seq - need);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
                acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][nf], af[ks][mf], acc[mf][nf], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // (activation fragment of the next K-step first: group (1, 0) then waits for all but the weight read behind it)
            if (ks == 0 && nf >= 10 - MF) AV_RA(1, nf - (10 - MF));
            if (nf + 2 < 10) {
                AV_RB(ks, nf + 2);
            } else if (ks == 0) {
                AV_RB(1, nf + 2 - 10);
            }
            if (npiece < MF + 5) piece(npiece++);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef AV_RA
#undef AV_RB
}

// RES: the launch adds a residual (p.R != nullptr).  A separate instantiation: its epilogue holds the residual rows of the whole
// wave tile in registers (requested right after the K loop, so that they land under the settle wait and the barrier that follow,
// and the epilogue itself issues no load at all -- a load there makes hipcc wait for the stores of the slabs before it).
template <int MF, bool GEGLU, int MODE, bool TRACE = false, bool SPLIT = false, bool RES = false>
__global__ __launch_bounds__(512) void gemm_big_kernel(const GemmK p) {
    static_assert(!(RES && (GEGLU || SPLIT)), "no residual on GEGLU / split-K launches");
    constexpr int BM = 64 * MF, BN = 320;  // four wave rows of MF 16-row fragments
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int SLAB_LD = (GEGLU ? 80 : 160) + 8;          // halves; 16-byte aligned rows
    constexpr int SLAB_BYTES = 16 * SLAB_LD * 2;             // per wave
    static_assert(8 * SLAB_BYTES <= STAGE_BYTES, "epilogue slabs must fit in one pipeline stage");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int G = gridDim.x;
    // Tile order.  Classic: output tiles N-fastest, dealt to the XCDs in contiguous runs of G / 8 (an XCD's 32 blocks then share
    // A panels in its L2).  Rastered (p.rast_gm > 0; wide-N launches, G = 256): the 32 concurrent blocks of XCD x (= blockIdx & 7)
    // cover ONE super-tile of rast_gm x rast_gn output tiles, and an XCD's consecutive super-tiles keep the same W slabs -- with
    // N = 16 / 32 tiles the classic order makes every XCD stream the whole W (6.6 / 26 MB > its 4 MB L2) once per round.
    const bool rast = !SPLIT && p.rast_gm > 0;
    const int b0 = rast ? (int)blockIdx.x : (((G & 7) == 0) ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x);
    const int tilesM = (p.M + BM - 1) / BM;
    const int ntiles_out = tilesM * p.tilesN;
    // SPLIT: work items are (split, output tile) -- split-K for launches whose tiles alone cannot fill the CUs (a separate
    // instantiation: the extra per-item state costs registers the plain kernel does not have)
    const int ntiles = SPLIT ? ntiles_out * p.splits : (rast ? G * ((p.rast_sm * p.rast_sn + 7) >> 3) : ntiles_out);
    // item -> output tile; false = a hole of the rastered order (ragged M, or past the last super-tile)
    auto decode = [&](int t, int& mt, int& nt) -> bool {
        if (!rast) {
            const int to = SPLIT ? t % ntiles_out : t;
            mt = to / p.tilesN;
            nt = to - mt * p.tilesN;
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

    const int srow0 = tid >> 3, pc = tid & 7, kc = pc ^ (srow0 & 7);
    const int ntap = p.nt0 + p.nt1;
    const int nk_all = p.taps * ntap;
    auto k_begin = [&](int item) { return SPLIT ? (nk_all * (item / ntiles_out)) / p.splits : 0; };
    auto k_end = [&](int item) { return SPLIT ? (nk_all * (item / ntiles_out + 1)) / p.splits : nk_all; };

    // ---- producer state (the tile whose K-tiles are being requested; runs ahead of the consumer by one K-tile) ----
    RowInfo ri[4];  // (entries >= MF unused)
    const half_t* bptr;               // W row (n_blk + srow0); the other four rows sit 64 * Ktot halves apart
    const size_t brow = (size_t)64 * p.Ktot;
    AGen<MODE> gen;
    auto producer_start = [&](int item) {
        int mt, nt;
        decode(item, mt, nt);
        const int kb = k_begin(item);
#pragma unroll
        for (int i = 0; i < 4; ++i) ri[i] = make_row<MODE>(p, i < MF ? mt * BM + srow0 + 64 * i : p.M);
        bptr = p.W + (size_t)(nt * BN + srow0) * p.Ktot + kc * 8 + (size_t)kb * 64;
        if constexpr (SPLIT)
            gen.start(p, ri, kc, kb, ntap);
        else
            gen.start(p, ri, kc);
    };
    auto advance = [&]() {
        bptr += 64;
        gen.next(p, ri, kc, ntap);
    };
    auto issue = [&](int stage) {
        char* st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < MF; ++i) glds16(gen.ap[i], st + (i * 512 + w * 64) * 16);
#pragma unroll
        for (int i = 0; i < 5; ++i) glds16(bptr + i * brow, st + A_BYTES + (i * 512 + w * 64) * 16);
        advance();
    };

    f4 acc[MF][10];
    int tile = next_valid(b0);
    if (tile >= ntiles) return;
    producer_start(tile);
    issue(0);
    int stage = 0;
    bool landed = false;
    bool rederive = false;  // producer state is not carried across an epilogue (register pressure): re-derive it  // the current tile's first K-tile was already waited for (before the previous epilogue)
    while (true) {
        int mt, nt;
        decode(tile, mt, nt);
        const int nk = k_end(tile) - k_begin(tile);
        const int m_wave = mt * BM + wr * MF * 16;
        const int n_wave = nt * BN + wc * 160;
        const int next_tile = next_valid(tile + G);
        const bool has_next = next_tile < ntiles;
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < 10; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        if (rederive) {
            producer_start(tile);
            advance();  // K-tile 0 of this tile was requested during the previous tile's last K-tile
        }

        for (int kt = 0; kt < nk; ++kt) {
            if constexpr (TRACE) if (tid == 0 && tile == b0 + AV_TRACE_TILE * G && kt < 8) p.trace[(size_t)blockIdx.x * 32 + 2 + 3 * kt] = (long long)__builtin_amdgcn_s_memtime();
            if (kt > 0 || !landed) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (TRACE) if (tid == 0 && tile == b0 + AV_TRACE_TILE * G && kt == 3) p.trace[(size_t)blockIdx.x * 32 + 29] = (long long)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();  // K-tile kt landed for everyone; everyone is done with the other stage
            if constexpr (TRACE) if (tid == 0 && tile == b0 + AV_TRACE_TILE * G && kt < 8) p.trace[(size_t)blockIdx.x * 32 + 3 + 3 * kt] = (long long)__builtin_amdgcn_s_memtime();
            const bool last = kt + 1 == nk;
            // the pieces below then fetch K-tile 0 of the next tile -- or, when the block has none, K-tile 0 of THIS tile again into the idle
            // stage (never read): the producer always addresses data that exists, so no piece needs a per-lane "fetch ? pointer : zero
            // line" select (16 v_cndmask per wave and K-tile out of the MFMA stream)
            if (last) producer_start(has_next ? next_tile : tile);
            const bool fetch = !last || has_next;
            const char* as = smem + stage * STAGE_BYTES;
            char* st = smem + (stage ^ 1) * STAGE_BYTES;
            mma_tile_big<MF>(acc, as, as + A_BYTES, wr, wc, lane, [&](int i) {
                if (i < MF)
                    glds16(gen.ap[i], st + (i * 512 + w * 64) * 16);
                else
                    glds16(bptr + (i - MF) * brow, st + A_BYTES + ((i - MF) * 512 + w * 64) * 16);
            });
            if constexpr (TRACE) if (tid == 0 && tile == b0 + AV_TRACE_TILE * G && kt < 8) p.trace[(size_t)blockIdx.x * 32 + 4 + 3 * kt] = (long long)__builtin_amdgcn_s_memtime();
            if (fetch) advance();
            stage ^= 1;
        }
        if constexpr (TRACE) if (tid == 0 && tile == b0 + AV_TRACE_TILE * G) p.trace[(size_t)blockIdx.x * 32 + 26] = (long long)__builtin_amdgcn_s_memtime();
        // `stage` now names the buffer holding the prefetched K-tile 0 of the next tile; stage ^ 1 was just consumed
        constexpr int OUT_W = GEGLU ? 80 : 160;       // output columns of this wave
        constexpr int CPRW = OUT_W / 8;               // 16-byte chunks per slab row
        constexpr int NIT = (16 * CPRW + 63) / 64;    // store iterations per slab (5, or 3 with a half-empty last one)
        const int n_out_wave = GEGLU ? n_wave / 2 : n_wave;
        h8 rr[RES ? MF : 1][NIT];
        if constexpr (RES) {   // all residual rows of the wave tile (rows past M: clamped, never stored)
            int lane_r = lane;
            asm volatile("" : "+v"(lane_r));   // (not hoisted out of the tile loop: see lane_e below)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int c = it * 64 + lane_r;
                    const int row = c / CPRW, cc = c - row * CPRW;
                    int m = m_wave + mf * 16 + row;
                    m = m < p.M ? m : p.M - 1;
                    rr[mf][it] = *(const h8*)(p.R + (size_t)m * p.ldr + n_out_wave + cc * 8);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // settle the prefetch BEFORE the stores below enter the queue
        if constexpr (TRACE) if (tid == 0 && tile == b0 + AV_TRACE_TILE * G) p.trace[(size_t)blockIdx.x * 32 + 30] = (long long)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();                     // every wave is done reading the consumed stage
        if constexpr (TRACE) if (tid == 0 && tile == b0 + AV_TRACE_TILE * G) p.trace[(size_t)blockIdx.x * 32 + 31] = (long long)__builtin_amdgcn_s_memtime();
        landed = true;
        rederive = true;

        // ---------------- wave-private epilogue: MF slabs of 16 rows x 160 (GEGLU: 80) output columns ----------------
        // (measured alternatives, both bit-equal and slower: pair-wise LDS exchange for full-row stores, round 1; a block-cooperative
        //  form -- four barrier-separated steps through LDS, all 512 threads storing whole rows -- round 2, 5-40 % slower on the
        //  3-clip shapes: with one block per CU nothing overlaps its serial steps.  profiles/r02_gemm_coop_epilogue_ab.txt)
        // the epilogue's lane-derived offsets must not be hoisted out of the tile loop (they would live across the K loop
        // and spill): launder the lane id once per tile
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int l15 = lane_e & 15, lq = lane_e >> 4;
        if constexpr (SPLIT) {  // raw fp32 partial tile; gemm_splitk_reduce_kernel sums the splits in order and finishes
            float* dst = p.partial + (size_t)(tile / ntiles_out) * p.M * p.N;
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int m = m_wave + mf * 16 + l15;
#pragma unroll
                for (int nf = 0; nf < 10; ++nf)
                    if (m < p.M) *(f4*)(dst + (size_t)m * p.N + n_wave + nf * 16 + 4 * lq) = acc[mf][nf];
            }
            if (!has_next) break;
            tile = next_tile;
            continue;
        }
        if constexpr (SPLIT) __builtin_unreachable();
        half_t* const slab = (half_t*)(smem + (stage ^ 1) * STAGE_BYTES + w * SLAB_BYTES);
        // (dispatch guarantees N % 320 == 0 and act in {none, GEGLU}; rows are guarded: M need not be a multiple of BM)
        h4 bvec[10];
#pragma unroll
        for (int nf = 0; nf < 10; ++nf)
            bvec[nf] = *(const h4*)(p.bias != nullptr ? p.bias + n_wave + nf * 16 + 4 * lq : p.zeros);
        if constexpr (RES) {   // they landed under the settle wait above; tell the compiler so ONCE, before the first store
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(rr[mf][it]));
        }
        // per 16-row slab: (+bias, +temb row vector | GEGLU) -> fp16 -> LDS (turns lane-owns-4-channels into
        // row-contiguous 16-byte chunks) -> (+residual) -> store.
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            if constexpr (GEGLU) {
#pragma unroll
                for (int np = 0; np < 5; ++np) {
                    h4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float hv = acc[mf][2 * np][r] + (float)bvec[2 * np][r];
                        const float gv = acc[mf][2 * np + 1][r] + (float)bvec[2 * np + 1][r];
                        o[r] = (half_t)(hv * av_gelu(gv));
                    }
                    *(h4*)(slab + l15 * SLAB_LD + np * 16 + 4 * lq) = o;
                }
            } else {
                const bool has_rv = p.rowvec != nullptr;
                const int mrow = m_wave + mf * 16 + l15;
                const half_t* rv = has_rv ? p.rowvec + (size_t)((mrow < p.M ? mrow : 0) / p.rowvec_div) * p.ldrv + n_wave + 4 * lq
                                          : p.zeros;
#pragma unroll
                for (int nf = 0; nf < 10; ++nf) {
                    h4 tv = (h4){0, 0, 0, 0};
                    if (has_rv) tv = *(const h4*)(rv + nf * 16);
                    h4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)(acc[mf][nf][r] + (float)bvec[nf][r] + (float)tv[r]);
                    *(h4*)(slab + l15 * SLAB_LD + nf * 16 + 4 * lq) = o;
                }
            }
            // same-wave LDS traffic is ordered; the compiler inserts the lgkmcnt wait for the read-back
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = it * 64 + lane_e;
                const int row = c / CPRW, cc = c - row * CPRW;
                const bool ok = (16 * CPRW % 64 == 0 || c < 16 * CPRW) && m_wave + mf * 16 + row < p.M;
                h8 v = *(const h8*)(slab + (ok ? row * SLAB_LD + cc * 8 : 0));
                if constexpr (RES) v = v + rr[mf][it];  // fp16 add: correctly rounded, == the fp32 add + rounding of two fp16 values
                if (ok) *(h8*)(p.C + (size_t)(m_wave + mf * 16 + row) * p.ldc + n_out_wave + cc * 8) = v;
            }
        }

        if constexpr (TRACE) {
            if (tid == 0 && tile == b0 + AV_TRACE_TILE * G) {
                p.trace[(size_t)blockIdx.x * 32 + 27] = (long long)__builtin_amdgcn_s_memtime();
                p.trace[(size_t)blockIdx.x * 32 + 28] = nk;
            }
        }
        if (!has_next) break;
        tile = next_tile;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Ping-pong persistent kernel (round 4): the long-K workhorse (3x3 / temporal convolutions, FF down-projections).
//
// Same block tile family as gemm_big_kernel (64 MF x 320 x 64, 8 waves 4 x 2, two LDS stages filled by LDS-DMA, persistent blocks,
// cross-tile prefetch, wave-private epilogue), different K-tile: gemm_big_kernel runs both waves of a SIMD through ONE schedule --
// they issue their LDS-DMA pieces, stall on them and want the matrix pipe at the same moments, and a K-tile costs ~3.5 k cycles
// against 1.9 k of MFMA work (profiles/r01_gemm_big_trace.txt, r02_gemm_dma_phase_experiment.txt).  Here a K-tile is four PHASES
// (k-step x column half), each a read slot R (this phase's fragments by ds_read_b128, a share of the next K-tile's LDS-DMA pieces,
// lgkmcnt(0)) and a matrix slot M (5 MF MFMAs back to back under s_setprio 1), every slot closed by s_barrier -- and waves 4-7 (the
// SIMD partners of waves 0-3) enter the loop ONE BARRIER LATE.  From then on a SIMD always has one wave in an M slot and its partner
// in the R slot of the following phase: the matrix pipe sees MFMA blocks back to back while all LDS / DMA issue happens beside them
// (MI355X_MICROARCH.md "Two waves per SIMD"; cdna_hip_programming.md 5, the 8-phase template's `if (wr == 1) s_barrier`).  The code
// is the same for both halves -- the offset is a barrier count, not a second schedule -- so hipcc sees one straight-line K-tile.
//
// Ordering (b = barrier index as waves 0-3 count them; waves 4-7 execute slot s between barriers s and s + 1):
//  * RAW, LDS-DMA -> ds_read: the pieces of K-tile kt + 1 are issued in R0 / R1 of K-tile kt and waited for (vmcnt(0)) at the end
//    of R3 of K-tile kt, BEFORE that slot's barrier, by every issuing wave; the first read of K-tile kt + 1 sits behind at least one
//    more barrier for every reader (R0 of waves 0-3 follows their M3; waves 4-7 run later still).
//  * WAR, ds_read -> LDS-DMA: every R slot ends with lgkmcnt(0) before its barrier; the stage that held K-tile kt - 1 is restaged from
//    R0 of K-tile kt on, i.e. behind the barrier that closed the last R3 of K-tile kt - 1 (waves 4-7) -- all its reads have returned.
//  * tile switch: waves 0-3 wait one extra barrier (until waves 4-7 are through their last M slot), both halves run their
//    wave-private epilogues through the consumed stage in the same interval, one barrier, then waves 4-7 fall back by one slot
//    again.  The next tile's first K-tile was requested during the last K-tile as usual and lies in the other stage.
template <int MF, int MODE, bool RES>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmK p) {
    constexpr int BM = 64 * MF, BN = 320;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int SLAB_LD = 160 + 8;                          // halves; 16-byte aligned rows
    constexpr int SLAB_BYTES = 16 * SLAB_LD * 2;              // per wave
    static_assert(8 * SLAB_BYTES <= STAGE_BYTES, "epilogue slabs must fit in one pipeline stage");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int grp = __builtin_amdgcn_readfirstlane(w >> 2);   // 0: waves 0-3 (lead), 1: waves 4-7 (one slot behind); an SGPR
    const int G = gridDim.x;
    const int b0 = ((G & 7) == 0) ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;  // XCD-contiguous
    const int tilesM = (p.M + BM - 1) / BM;
    const int ntiles = tilesM * p.tilesN;
    const int srow0 = tid >> 3, pc = tid & 7, kc = pc ^ (srow0 & 7);
    const int ntap = p.nt0 + p.nt1;
    const int nk = p.taps * ntap;

    // ---- producer state (the K-tile being requested: one ahead of the one being multiplied) ----
    RowInfo ri[4];
    const half_t* bptr;
    const size_t brow = (size_t)64 * p.Ktot;
    AGen<MODE> gen;
    auto producer_start = [&](int item) {
        const int mt = item / p.tilesN, nt = item - mt * p.tilesN;
#pragma unroll
        for (int i = 0; i < 4; ++i) ri[i] = make_row<MODE>(p, i < MF ? mt * BM + srow0 + 64 * i : p.M);
        bptr = p.W + (size_t)(nt * BN + srow0) * p.Ktot + kc * 8;
        gen.start(p, ri, kc);
    };
    auto advance = [&]() {
        bptr += 64;
        gen.next(p, ri, kc, ntap);
    };
    auto piece = [&](int i, char* st) {   // i: constant after unrolling
        if (i < MF)
            glds16(gen.ap[i], st + (i * 512 + w * 64) * 16);
        else
            glds16(bptr + (i - MF) * brow, st + A_BYTES + ((i - MF) * 512 + w * 64) * 16);
    };

    int tile = b0;
    if (tile >= ntiles) return;
    producer_start(tile);
#pragma unroll
    for (int i = 0; i < MF + 5; ++i) piece(i, smem);
    advance();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __builtin_amdgcn_s_barrier();         // K-tile 0 of the first tile is in stage 0 for everyone
    if (grp == 1) __builtin_amdgcn_s_barrier();   // the stagger: waves 4-7 start one slot late
    int stage = 0;
    bool rederive = false;
    f4 acc[MF][10];

    const int l15 = lane & 15, lq = lane >> 4;
    const int c0 = ((0 * 4 + lq) ^ (l15 & 7)) * 16, c1 = ((1 * 4 + lq) ^ (l15 & 7)) * 16;
    const int a_off = (wr * MF * 16 + l15) * 128, b_off = A_BYTES + (wc * 160 + l15) * 128;

    while (true) {
        const int mt = tile / p.tilesN, nt = tile - mt * p.tilesN;
        const int m_wave = mt * BM + wr * MF * 16;
        const int n_wave = nt * BN + wc * 160;
        const int next_tile = tile + G;
        const bool has_next = next_tile < ntiles;
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < 10; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        if (rederive) {
            producer_start(tile);
            advance();  // K-tile 0 of this tile was requested during the previous tile's last K-tile
        }

        for (int kt = 0; kt < nk; ++kt) {
            const bool last = kt + 1 == nk;
            if (last && has_next) producer_start(next_tile);  // the pieces below then fetch K-tile 0 of the next tile
            const bool fetch = !last || has_next;
            const unsigned sb = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(smem + stage * STAGE_BYTES);
            char* st = smem + (stage ^ 1) * STAGE_BYTES;
            const unsigned abase[2] = {sb + a_off + c0, sb + a_off + c1};
            const unsigned bbase[2] = {sb + b_off + c0, sb + b_off + c1};
            h8 af[MF], bf[5];
#define AV_PP_SLOT_END()                           \
    __builtin_amdgcn_sched_barrier(0);             \
    __builtin_amdgcn_s_barrier();                  \
    __builtin_amdgcn_sched_barrier(0)
#define AV_PP_READS(KS, NH, WITH_A)                                                              \
    if (WITH_A) {                                                                                \
        _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) af[mf] = lds_frag(abase[KS], mf * 2048); \
    }                                                                                            \
    _Pragma("unroll") for (int nf = 0; nf < 5; ++nf) bf[nf] = lds_frag(bbase[KS], ((NH) * 5 + nf) * 2048)
#define AV_PP_MFMAS(NH)                                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                          \
    AV_PP_SLOT_END();                                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                                              \
    _Pragma("unroll") for (int mf = 0; mf < MF; ++mf)                                                                           \
        _Pragma("unroll") for (int nf = 0; nf < 5; ++nf)                                                                        \
            acc[mf][(NH) * 5 + nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nf], af[mf], acc[mf][(NH) * 5 + nf], 0, 0, 0);   \
    __builtin_amdgcn_s_setprio(0);                                                                                              \
    AV_PP_SLOT_END()
            // (in every R slot the DMA issue / address arithmetic comes FIRST and the asm fragment reads last, directly in front of
            //  their wait: nothing that needs registers may sit between an asm read and its lgkmcnt -- hipcc would be free to spill
            //  a destination that has not arrived yet, tests/test_isa_guards.py)
            // ---- phase 0: k-step 0, columns 0..79 of the wave tile; pieces A0 .. A(MF-1), W0 of the next K-tile
            if (fetch) {
#pragma unroll
                for (int i = 0; i < MF + 1; ++i) piece(i, st);
            }
            __builtin_amdgcn_sched_barrier(0);
            AV_PP_READS(0, 0, true);
            AV_PP_MFMAS(0);
            // ---- phase 1: k-step 0, columns 80..159; pieces W1 .. W4
            if (fetch) {
#pragma unroll
                for (int i = MF + 1; i < MF + 5; ++i) piece(i, st);
                advance();
            }
            __builtin_amdgcn_sched_barrier(0);
            AV_PP_READS(0, 1, false);
            AV_PP_MFMAS(1);
            // ---- phase 2: k-step 1, columns 0..79
            AV_PP_READS(1, 0, true);
            AV_PP_MFMAS(0);
            // ---- phase 3: k-step 1, columns 80..159; the next K-tile has landed (this wave's pieces) before the slot's barrier
            AV_PP_READS(1, 1, false);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            AV_PP_MFMAS(1);
#undef AV_PP_READS
#undef AV_PP_MFMAS
            stage ^= 1;
        }
        // `stage` now names the buffer holding the prefetched K-tile 0 of the next tile; stage ^ 1 was just consumed
        constexpr int CPRW = 160 / 8;                 // 16-byte chunks per slab row
        constexpr int NIT = (16 * CPRW + 63) / 64;    // store iterations per slab (5)
        h8 rr[RES ? 2 : 1][NIT];
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));              // (keeps the epilogue's address math out of the K loop's live ranges)
        auto load_res = [&](int mf) {                 // residual rows of slab mf (rows past M: clamped, never stored)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = it * 64 + lane_r;
                const int row = c / CPRW, cc = c - row * CPRW;
                int m = m_wave + mf * 16 + row;
                m = m < p.M ? m : p.M - 1;
                rr[mf & 1][it] = *(const h8*)(p.R + (size_t)m * p.ldr + n_wave + cc * 8);
            }
        };
        if constexpr (RES) {
            load_res(0);
            load_res(1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0) __builtin_amdgcn_s_barrier();   // waves 4-7 are in their last M slot; their last reads were waited for two barriers ago
        __builtin_amdgcn_sched_barrier(0);
        rederive = true;

        // ---------------- wave-private epilogue: MF slabs of 16 rows x 160 output columns through the consumed stage ----------------
        const int l15e = lane_r & 15, lqe = lane_r >> 4;
        half_t* const slab = (half_t*)(smem + (stage ^ 1) * STAGE_BYTES + w * SLAB_BYTES);
        h4 bvec[10];
#pragma unroll
        for (int nf = 0; nf < 10; ++nf)
            bvec[nf] = *(const h4*)(p.bias != nullptr ? p.bias + n_wave + nf * 16 + 4 * lqe : p.zeros);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const bool has_rv = p.rowvec != nullptr;
            const int mrow = m_wave + mf * 16 + l15e;
            const half_t* rv = has_rv ? p.rowvec + (size_t)((mrow < p.M ? mrow : 0) / p.rowvec_div) * p.ldrv + n_wave + 4 * lqe : p.zeros;
#pragma unroll
            for (int nf = 0; nf < 10; ++nf) {
                h4 tv = (h4){0, 0, 0, 0};
                if (has_rv) tv = *(const h4*)(rv + nf * 16);
                h4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (half_t)(acc[mf][nf][r] + (float)bvec[nf][r] + (float)tv[r]);
                *(h4*)(slab + l15e * SLAB_LD + nf * 16 + 4 * lqe) = o;
            }
            h8 v[NIT];
            bool ok[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = it * 64 + lane_r;
                const int row = c / CPRW, cc = c - row * CPRW;
                ok[it] = m_wave + mf * 16 + row < p.M;
                v[it] = *(const h8*)(slab + row * SLAB_LD + cc * 8);
                if constexpr (RES) v[it] = v[it] + rr[mf & 1][it];  // fp16 add: correctly rounded, == the fp32 add + rounding of two fp16 values
            }
            if constexpr (RES) {
                if (mf + 2 < MF) load_res(mf + 2);   // requested BEFORE this slab's stores: its wait will not have to drain them
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = it * 64 + lane_r;
                const int row = c / CPRW, cc = c - row * CPRW;
                if (ok[it]) *(h8*)(p.C + (size_t)(m_wave + mf * 16 + row) * p.ldc + n_wave + cc * 8) = v[it];
            }
        }
        if (!has_next) break;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the slab read-backs feed the stores above; belt and braces)
        __builtin_amdgcn_s_barrier();                  // both halves are done with the consumed stage: it may be restaged
        if (grp == 1) __builtin_amdgcn_s_barrier();    // waves 4-7 fall one slot behind again
        __builtin_amdgcn_sched_barrier(0);
        tile = next_tile;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Reference-grade kernel: one thread per output element, any shape.  Used for the tiny once-per-clip
// conditioning layers (Cin = 4/16/32 ...) and as the on-device cross-check of the MFMA kernels in the tests.
template <int MODE>
__global__ void gemm_naive_kernel(const GemmK p) {
    const bool geglu = p.act == ACT_GEGLU;
    const int Nout = geglu ? p.N / 2 : p.N;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)p.M * Nout) return;
    const int m = (int)(idx / Nout), j = (int)(idx - (long long)m * Nout);
    const int n = geglu ? 32 * (j / 16) + (j % 16) : j;
    const RowInfo ri = make_row<MODE>(p, m);
    const int K = p.C0 + p.C1;
    float a0 = 0.f, a1 = 0.f;
    for (int tap = 0; tap < p.taps; ++tap) {
        const int sr = src_row<MODE>(p, ri, tap);
        if (sr < 0) continue;
        const half_t* w0 = p.W + (size_t)n * p.Ktot + (size_t)tap * K;
        const half_t* w1 = w0 + (size_t)16 * p.Ktot;
        const half_t* x0 = p.A0 + (size_t)sr * p.lda0;
        for (int k = 0; k < p.C0; ++k) {
            const float x = (float)x0[k];
            a0 += x * (float)w0[k];
            if (geglu) a1 += x * (float)w1[k];
        }
        if (p.C1 > 0) {
            const half_t* x1 = p.A1 + (size_t)sr * p.lda1;
            for (int k = 0; k < p.C1; ++k) {
                const float x = (float)x1[k];
                a0 += x * (float)w0[p.C0 + k];
                if (geglu) a1 += x * (float)w1[p.C0 + k];
            }
        }
    }
    float v;
    if (geglu) {
        if (p.bias != nullptr) {
            a0 += (float)p.bias[n];
            a1 += (float)p.bias[n + 16];
        }
        v = a0 * av_gelu(a1);
    } else {
        v = a0;
        if (p.bias != nullptr) v += (float)p.bias[n];
        if (p.rowvec != nullptr) v += (float)p.rowvec[(size_t)(m / p.rowvec_div) * p.ldrv + n];
        if (p.act == ACT_SILU)
            v = av_silu(v);
        else if (p.act == ACT_GELU)
            v = av_gelu(v);
    }
    if (p.act == ACT_F32OUT) {
        ((float*)p.C)[(size_t)m * p.ldc + j] = v;
        return;
    }
    if (p.R != nullptr) v = (float)(half_t)v + (float)p.R[(size_t)m * p.ldr + j];
    p.C[(size_t)m * p.ldc + j] = (half_t)v;
}

// split-K second pass: sum the fp32 partial tiles in a fixed order (deterministic), then the usual epilogue
__global__ void gemm_splitk_reduce_kernel(const GemmK p) {
    const int N8 = p.N >> 3;
    const long long total = (long long)p.M * N8;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / N8), n0 = (int)(idx - (long long)m * N8) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        for (int s = 0; s < p.splits; ++s) {
            const float* src = p.partial + ((size_t)s * p.M + m) * p.N + n0;
            const f4 a = *(const f4*)src, b = *(const f4*)(src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += a[e];
                v[4 + e] += b[e];
            }
        }
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = v[e];
            if (p.bias != nullptr) x += (float)p.bias[n0 + e];
            if (p.rowvec != nullptr) x += (float)p.rowvec[(size_t)(m / p.rowvec_div) * p.ldrv + n0 + e];
            if (p.act == ACT_SILU)
                x = av_silu(x);
            else if (p.act == ACT_GELU)
                x = av_gelu(x);
            o[e] = (half_t)x;
        }
        if (p.R != nullptr) {
            const h8 rr = *(const h8*)(p.R + (size_t)m * p.ldr + n0);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)o[e] + (float)rr[e]);
        }
        *(h8*)(p.C + (size_t)m * p.ldc + n0) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------
static const half_t* zero_line() {
    static const half_t* z = nullptr;
    if (z == nullptr) {
        void* ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_zero_line)) == hipSuccess) z = (const half_t*)ptr;
    }
    return z;
}

// K-tiles (of 64) from which the ping-pong kernel would be taken by default.  Measured (profiles/r04_gemm_pp_ab_v1_*.txt, interleaved A/B
// on the edit step's launches): its K loop is 2-6 % faster than gemm_big_kernel's from K = 5760 on (conv 960->320 @64x64 898 -> 845 us,
// 1.21 -> 1.29 PF), equal at K = 2880, and its tile switch costs more (residual launches 10-30 % slower; temporal convolutions,
// FF-down slower) -- the R slots, not the M slots, set the slot time (9 LDS-DMA issues per wave and K-tile).  The launches it wins sum
// to 0.25 ms of the 106 ms step pair (+ 0.08 ms on the inversion step's one-round launches, M = 65536 with 256-row tiles: conv 960->320
// 322 -> 298 us, r04_gemm_pp_ab_v1_b1_*.txt), so it stays OFF by default (flags bit17 selects it: tests, A/B); a second form with the next
// tile's start-up hoisted in front of the tile-switch barrier was slower throughout (r04_gemm_pp_ab_v2_*.txt, not kept).
constexpr int AV_PP_MIN_KTILES = 1 << 30;

template <int MODE>
static int dispatch(GemmK& k, const AnyV2VGemmDesc* d, bool fast, hipStream_t s) {
    const bool geglu = d->act == ACT_GEGLU;
    if (!fast) {
        const int Nout = geglu ? d->N / 2 : d->N;
        const long long total = (long long)d->M * Nout;
        hipLaunchKernelGGL(gemm_naive_kernel<MODE>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k);
        return av_launch_status("gemm_naive");
    }
    const bool glds = (d->flags & 2) != 0;
    // Stream-K form of the one-wave-per-SIMD kernel (gemm_sw.hip): flags bit26 allows it where av_gemm_sw_sk_blocks says it pays,
    // bit27 forces it; never for a batch-hinted launch (its K ranges depend on the launch's own tile count, i.e. they fix the
    // arithmetic, and a hinted launch has to reproduce the arithmetic of the launch it stands for).
    if (glds && (d->flags & ((1 << 26) | (1 << 27))) && !(d->flags & (1 << 22)) && av_gemm_sw_eligible(d) && av_hint_rows(d->M) == d->M &&
        d->workspace != nullptr) {
        const int blocks = av_gemm_sw_sk_blocks(d, (d->flags & (1 << 27)) != 0);
        if (blocks > 0 && av_gemm_sw_sk_workspace(blocks) <= (size_t)d->workspace_bytes) return av_gemm_sw_sk_launch(k, d, blocks, s);
    }
    // 3x3 convolution with LDS reuse of the A operand across the dx taps (gemm_swh.hip): flags bit28 takes it where eligible.
    if (glds && (d->flags & (1 << 28)) && !(d->flags & (1 << 22)) && av_gemm_swh_eligible(d)) return av_gemm_swh_launch(k, d, s);
    // One-wave-per-SIMD persistent kernel (gemm_sw.hip): flags bit21 takes it wherever the shape allows, bit22 forbids it.
    if (glds && (d->flags & (1 << 21)) && !(d->flags & (1 << 22)) && av_gemm_sw_eligible(d)) return av_gemm_sw_launch(k, d, s);
    // 128-row kernel tile width: 160 columns (NF = 5) where N allows it, except where 128-column tiles (NF = 4) quantise better onto
    // the 256 CUs x 2 resident blocks -- more CUs busy when there is less than one tile per CU, or the same number of rounds with
    // 20 % smaller tiles (flags bit11 / bit12 force NF = 4 / 5: A/B in tools/gemm_nf_ab.py).  Same arithmetic per output either way.
    // The width does not touch the arithmetic, so a launch picks it on its OWN rows; the split-K factor of a batch-hinted launch is
    // the reference launch's, i.e. planned with the width the reference launch picks (nf_ref below).
    const int nk_all = k.taps * (k.nt0 + k.nt1);
    auto choose_nf = [&](int rows) -> int {
        int nf_ = geglu ? 4 : (d->N % 160 == 0 ? 5 : 4);
        if (!geglu && nf_ == 5 && d->N % 128 == 0) {
            const int mt = (rows + 127) / 128;
            const int t5 = mt * (d->N / 160), t4 = mt * (d->N / 128);
            // (not where the launch would be split along K: the split factor is derived from the tile count, and 7 x 80 tiles spill
            //  into a second round where 7 x 64 do not -- B = 1 8x8-level convolutions: 45 -> 59 us, profiles/r03_gemm_nf_ab.txt)
            const bool would_split = t4 < 384 && ((t4 <= 128 && nk_all >= 32) || nk_all >= 72);
            const bool prefer4 = !would_split && ((t4 <= 256) || (t5 > 256 && (t5 + 511) / 512 == (t4 + 511) / 512));
            if (((d->flags & 2048) || prefer4) && !(d->flags & 4096)) nf_ = 4;
        }
        return nf_;
    };
    const int nf = choose_nf(d->M);
    const int nf_ref = choose_nf(av_hint_rows(d->M));
    const int tilesN_small = (d->N + nf * 32 - 1) / (nf * 32);
    constexpr int BMB = 192;
    const bool big_ok = glds && !(d->flags & 4) && d->N % 320 == 0 && (!geglu || MODE == MODE_LINEAR) && (geglu || d->act == ACT_NONE);
    // Launch plan as a function of the row count: kernel family (persistent 192 x 320 tiles / 128-row tiles) and split-K factor.
    //  * persistent kernel: taken when its tiles fill the 256 CUs for a whole number of rounds well enough (>= 75 %), or when
    //    forced (flags bit3);
    //  * launches that cannot fill the CUs but have a long K loop (8x8-level convs / FF-down of the 3-clip batch, M = 3072): split K
    //    so that (tiles x splits) is one nearly full round of 256 work items; the ordered reduce pass finishes them.  Measured
    //    (profiles/r01_gemm_split_ab.txt): 1.2-1.4x over the 128-row kernel's split path from 80 K-tiles on with >= 224 work items;
    //    slower below 72 K-tiles or with a 3/4-full round (M = 1024), which stay on the 128-row kernel;
    //  * 128-row kernel split-K for launches that cannot fill the chip (512 block slots) and have a long K loop (with 20 K-tiles the
    //    second pass costs more than the idle CUs; with 60 it pays only when fewer than a quarter of the block slots would be busy;
    //    from ~72 K-tiles on it always pays).
    // (the workspace test uses the PLANNED row count as well: a batch-hinted launch must reproduce the decision of the launch it
    //  stands for -- its own, smaller partial tiles could fit where the reference launch's do not, and the two would then split
    //  differently: seen at 16 f x 256^2, tests/test_gpu_parity.py::test_two_branch_steps_bit_equal_at_a_mid_size_full_width)
    // (the split-K rules below see at most the 64 MiB the workspace had when they were tuned: a larger buffer -- the stream-K form
    //  wants 126 MB -- must not change which launches split, i.e. their arithmetic)
    const size_t split_ws_bytes = (size_t)d->workspace_bytes < ((size_t)64 << 20) ? (size_t)d->workspace_bytes : ((size_t)64 << 20);
    struct Plan { int big, splits; };
    auto plan = [&](int rows, int nf_rows) -> Plan {
        const bool ws_ok = d->workspace != nullptr && d->N % 8 == 0;
        if (big_ok) {
            const int tb = ((rows + BMB - 1) / BMB) * (d->N / 320);
            const int rounds = (tb + 255) / 256;
            const bool fills = tb >= 224 && tb * 4 >= rounds * 256 * 3;
            if (!fills && !geglu && !(d->flags & (16 | 8)) && ws_ok && tb <= 128 && nk_all >= 72) {
                int sp = 256 / tb;
                if (sp > 8) sp = 8;
                if (sp > nk_all / 12) sp = nk_all / 12;
                if (sp >= 2 && tb * sp >= 224 && (size_t)sp * rows * d->N * sizeof(float) <= split_ws_bytes) return {1, sp};
            }
            if (fills || (d->flags & 8)) return {1, 1};
        }
        const int tm = ((rows + 127) / 128) * ((d->N + nf_rows * 32 - 1) / (nf_rows * 32));
        const bool split_pays = (tm <= 128 && nk_all >= 32) || nk_all >= 72;
        if (glds && !geglu && d->act != ACT_F32OUT && !(d->flags & 16) && ws_ok && tm < 384 && split_pays) {
            int sp = (512 + tm - 1) / tm;
            if (sp > 8) sp = 8;
            if (sp > nk_all / 8) sp = nk_all / 8;
            if (sp >= 2 && (size_t)sp * rows * d->N * sizeof(float) <= split_ws_bytes) return {0, sp};
        }
        return {0, 1};
    };
    // Batch hint (anyv2v_set_batch_hint): what fixes the ARITHMETIC is the split-K factor (fp32 partial tiles summed afterwards);
    // the two kernel families accumulate every output element in the same order (tests/gpu_checks.py asserts it bit for bit).  A
    // hinted launch therefore takes the split factor of the launch it stands for and is otherwise planned on its own row count.
    Plan use = plan(d->M, nf);
    if (av_hint_rows(d->M) != d->M) {
        const Plan ref = plan(av_hint_rows(d->M), nf_ref);
        if (ref.splits > 1)
            use = ref;
        else if (use.splits > 1)
            use = Plan{0, 1};
    }
    // Ping-pong kernel (gemm_pp_kernel): flags bit17 takes it wherever the shape allows, bit18 forbids it, bit19 / bit20 force its
    // 192- / 256-row tile (default: the taller tile unless it quantises worse onto the 256 CUs).  Not split along K (the launches
    // that want that are too small for it), so a batch-hinted launch may only take it when its reference launch is unsplit too.
    if (big_ok && !geglu && use.splits == 1 && !(d->flags & (1 << 18)) &&
        ((d->flags & (1 << 17)) || (use.big && nk_all >= AV_PP_MIN_KTILES))) {
        auto eff = [&](int bm) {
            const int tb = ((d->M + bm - 1) / bm) * (d->N / 320), r = (tb + 255) / 256;
            return (double)tb / (r * 256.0);
        };
        const int mf = (d->flags & (1 << 19)) ? 3 : ((d->flags & (1 << 20)) ? 4 : (eff(256) + 0.02 >= eff(192) ? 4 : 3));
        const int tb = ((d->M + 64 * mf - 1) / (64 * mf)) * (d->N / 320);
        const dim3 gridp(tb < 256 ? tb : 256);
        k.tilesN = d->N / 320;
#define AV_PP(MF_)                                                                                        \
    do {                                                                                                  \
        if (d->R != nullptr)                                                                              \
            hipLaunchKernelGGL((gemm_pp_kernel<MF_, MODE, true>), gridp, dim3(512), 0, s, k);             \
        else                                                                                              \
            hipLaunchKernelGGL((gemm_pp_kernel<MF_, MODE, false>), gridp, dim3(512), 0, s, k);            \
    } while (0)
        if (mf == 4)
            AV_PP(4);
        else
            AV_PP(3);
#undef AV_PP
        return av_launch_status("gemm_pp");
    }
    {   // ANYV2V_GEMM_LOG=1: one line per launch plan on stderr (diagnostics: which launches split, and how, under a batch hint)
        static const bool log_on = getenv("ANYV2V_GEMM_LOG") != nullptr;
        if (log_on)
            fprintf(stderr, "gemm-plan mode %d M %d (hinted %d) N %d K %d act %d res %d big %d splits %d\n", MODE, d->M, av_hint_rows(d->M), d->N,
                    k.Ktot, d->act, d->R != nullptr, use.big, use.splits);
    }
    if (use.big) {
        const int tiles_big = ((d->M + BMB - 1) / BMB) * (d->N / 320);
        k.tilesN = d->N / 320;
        if (use.splits > 1) {
            k.splits = use.splits;
            k.partial = (float*)d->workspace;
            const dim3 grid(tiles_big * use.splits < 256 ? tiles_big * use.splits : 256);
            hipLaunchKernelGGL((gemm_big_kernel<3, false, MODE, false, true>), grid, dim3(512), 0, s, k);
            const long long total = (long long)d->M * (d->N / 8);
            long long blocks = (total + 255) / 256;
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, k);
            return av_launch_status("gemm_big<split-K>");
        }
        const dim3 grid(tiles_big < 256 ? tiles_big : 256);
        {   // tile order of wide-N launches (gemm_big_kernel): 8 x 4 super-tiles per XCD round when N has >= 8 tiles (the GEGLU
            // up-projections at 640 / 1280 channels: 16 / 32 N-tiles).  flags bits 13-15: 0 auto, 1 classic order, 2..6 force
            // rast_gm = 4, 8, 16, 32, 2; bit16: super-tiles N-fastest.  Same arithmetic per output element in every order.
            const int code = (d->flags >> 13) & 7;
            static const int gm_of[8] = {0, 0, 4, 8, 16, 32, 2, 0};
            int gm = gm_of[code];
            if (code == 0 && k.tilesN >= 8 && tiles_big >= 512) gm = 8;
            const int tm_big = (d->M + BMB - 1) / BMB;
            if (gm > 0 && grid.x == 256 && k.tilesN % (32 / gm) == 0) {
                k.rast_gm = gm;
                k.rast_gn = 32 / gm;
                k.rast_sm = (tm_big + gm - 1) / gm;
                k.rast_sn = k.tilesN / k.rast_gn;
                k.rast_nfast = (d->flags >> 16) & 1;
            }
        }
#ifdef ANYV2V_EXPERIMENTS  // probe build only (make experiments): phase-timestamp instantiations, tools/gemm_big_trace.py
#include "../../tools/experiments/gemm_dispatch_big_probe.inc"
#endif
        if constexpr (MODE == MODE_LINEAR) {
            if (geglu)
                hipLaunchKernelGGL((gemm_big_kernel<3, true, MODE_LINEAR>), grid, dim3(512), 0, s, k);
            else if (d->R != nullptr)
                hipLaunchKernelGGL((gemm_big_kernel<3, false, MODE_LINEAR, false, false, true>), grid, dim3(512), 0, s, k);
            else
                hipLaunchKernelGGL((gemm_big_kernel<3, false, MODE_LINEAR>), grid, dim3(512), 0, s, k);
        } else if (d->R != nullptr) {
            hipLaunchKernelGGL((gemm_big_kernel<3, false, MODE, false, false, true>), grid, dim3(512), 0, s, k);
        } else {
            hipLaunchKernelGGL((gemm_big_kernel<3, false, MODE>), grid, dim3(512), 0, s, k);
        }
        return av_launch_status("gemm_big");
    }
    k.tilesN = tilesN_small;
    const int tiles = ((d->M + 127) / 128) * k.tilesN;
    const int nk = nk_all;
    (void)nk;
    if (use.splits > 1) {
        k.splits = use.splits;
        k.partial = (float*)d->workspace;
    }
    const dim3 grid(tiles * k.splits);
#ifdef ANYV2V_EXPERIMENTS  // probe build only: phase timestamps (flag 32) / K-loop knock-outs (flags 64..448), tools/gemm_trace.py
#include "../../tools/experiments/gemm_dispatch_mfma_probe.inc"
#endif
#define AV_LAUNCH2(NF_, GEGLU_)                                                                          \
    do {                                                                                                 \
        if (glds)                                                                                   \
            hipLaunchKernelGGL((gemm_mfma_kernel<NF_, true, GEGLU_, MODE>), grid, dim3(256), 0, s, k);   \
        else                                                                                             \
            hipLaunchKernelGGL((gemm_mfma_kernel<NF_, false, GEGLU_, MODE>), grid, dim3(256), 0, s, k);  \
    } while (0)
    if (geglu)
        AV_LAUNCH2(4, true);
    else if (nf == 5)
        AV_LAUNCH2(5, false);
    else
        AV_LAUNCH2(4, false);
#undef AV_LAUNCH2
    if (k.splits > 1) {
        const long long total = (long long)d->M * (d->N / 8);
        long long blocks = (total + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, k);
    }
    return av_launch_status("gemm_mfma");
}

extern "C" int anyv2v_gemm_f16(const AnyV2VGemmDesc* d, void* stream) {
    AV_CHECK(d != nullptr, "gemm: null descriptor");
    AV_CHECK(d->A0 && d->W && d->C, "gemm: null A0/W/C");
    AV_CHECK(d->M > 0 && d->N > 0 && d->C0 > 0 && d->C1 >= 0, "gemm: bad M/N/C0/C1 (%d %d %d %d)", d->M, d->N, d->C0, d->C1);
    AV_CHECK(d->mode >= 0 && d->mode <= 2, "gemm: bad mode %d", d->mode);
    AV_CHECK(d->act >= 0 && d->act <= 4, "gemm: bad act %d", d->act);
    AV_CHECK(d->act != ACT_F32OUT || (d->rowvec == nullptr && d->R == nullptr && d->N % 4 == 0 && d->ldc % 4 == 0),
             "gemm: fp32 output supports bias only and needs N, ldc multiples of 4");
    AV_CHECK(d->asym == 0 || d->asym == 1, "gemm: asym must be 0 or 1");
    AV_CHECK(d->C1 == 0 || d->A1 != nullptr, "gemm: C1 > 0 but A1 is null");
    AV_CHECK(d->rowvec == nullptr || d->rowvec_div > 0, "gemm: rowvec needs rowvec_div > 0");
    if (d->mode == MODE_CONV2D) {
        AV_CHECK(d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0 && (d->stride == 1 || d->stride == 2),
                 "gemm: bad conv geometry");
        AV_CHECK(d->M % (d->Ho * d->Wo) == 0, "gemm: M not a multiple of Ho*Wo");
        AV_CHECK(d->up == 0 || d->up == 1, "gemm: up must be 0 or 1");
    }
    if (d->mode == MODE_TEMPORAL) {
        AV_CHECK(d->F > 0 && d->HW > 0 && d->M % (d->F * d->HW) == 0, "gemm: bad temporal geometry");
    }
    if (d->act == ACT_GEGLU) {
        AV_CHECK(d->N % 32 == 0, "gemm: GEGLU needs N %% 32 == 0");
        AV_CHECK(d->rowvec == nullptr, "gemm: GEGLU with rowvec unsupported");
    }
    GemmK k;
    k.A0 = (const half_t*)d->A0;
    k.A1 = d->C1 > 0 ? (const half_t*)d->A1 : (const half_t*)d->A0;
    k.W = (const half_t*)d->W;
    k.C = (half_t*)d->C;
    k.bias = (const half_t*)d->bias;
    k.rowvec = (const half_t*)d->rowvec;
    k.R = (const half_t*)d->R;
    k.zeros = zero_line();
    AV_CHECK(k.zeros != nullptr, "gemm: zero line symbol unavailable");
    k.M = d->M; k.N = d->N; k.C0 = d->C0; k.C1 = d->C1;
    k.lda0 = d->lda0; k.lda1 = d->C1 > 0 ? d->lda1 : d->lda0; k.ldc = d->ldc; k.ldr = d->ldr; k.ldrv = d->ldrv;
    k.rowvec_div = d->rowvec_div > 0 ? d->rowvec_div : 1;
    k.mode = d->mode; k.Hi = d->Hi; k.Wi = d->Wi; k.Ho = d->Ho; k.Wo = d->Wo; k.stride = d->stride; k.up = d->up;
    k.F = d->F; k.HW = d->HW; k.act = d->act;
    k.pad_lo = d->asym ? 0 : 1;
    k.taps = d->mode == MODE_LINEAR ? 1 : (d->mode == MODE_CONV2D ? 9 : 3);
    k.Ktot = k.taps * (d->C0 + d->C1);
    k.nt0 = d->C0 / 64;
    k.nt1 = d->C1 / 64;
    k.tilesN = 1;
    k.splits = 1;
    k.partial = nullptr;
    k.trace = nullptr;
    k.ln_c1 = nullptr;
    k.ln_eps = 0.f;
    k.rast_gm = k.rast_gn = k.rast_sm = k.rast_sn = k.rast_nfast = 0;
    k.vec_epi = (((uintptr_t)d->bias & 7) == 0) && (((uintptr_t)d->rowvec & 7) == 0) && (d->ldrv % 4 == 0) && (d->N % 4 == 0);
    hipStream_t s = (hipStream_t)stream;

    const bool geglu = d->act == ACT_GEGLU;
    const bool fast = !(d->flags & 1) && d->C0 % 64 == 0 && d->C1 % 64 == 0 && d->lda0 % 8 == 0 &&
                      (d->C1 == 0 || d->lda1 % 8 == 0) && d->ldc % 8 == 0 && av_aligned16(d->A0) &&
                      (d->C1 == 0 || av_aligned16(d->A1)) && av_aligned16(d->W) && av_aligned16(d->C) &&
                      (d->R == nullptr || (d->ldr % 8 == 0 && av_aligned16(d->R))) && (!geglu || d->N % 128 == 0) &&
                      k.vec_epi;
    // K = 320 Linear layers with many rows (the 64x64 level): weight-stationary streaming kernel (gemm_ws.hip).
    // flags bit9 (512): never, bit10 (1024): whenever the shape allows (tests; small M leaves most waves idle)
    if (d->ln_c1 != nullptr) {   // LayerNorm folded into the GEMM: only the weight-stationary kernel implements it
        if (!(fast && (d->flags & 2) && !(d->flags & 1) && av_gemm_ws_eligible(d) && (((uintptr_t)d->ln_c1) & 15) == 0)) {
            anyv2v_set_error("gemm: ln_c1 (LayerNorm fold) needs mode 0, C0 = 320 (N %% 160 = 0) or C0 = 512 with GEGLU (N %% 128 = 0), no "
                             "residual / rowvec, 16-byte aligned operands -- got C0 %d N %d act %d", d->C0, d->N, d->act);
            return ANYV2V_EUNSUPPORTED;
        }
        return av_gemm_ws_launch(k, d, s);
    }
    if (fast && (d->flags & 2) && !(d->flags & (512 | 4 | 1)) && av_gemm_ws_eligible(d) &&
        (av_hint_rows(d->M) >= 32768 || (d->flags & 1024)))
        return av_gemm_ws_launch(k, d, s);
    if (d->mode == MODE_CONV2D) return dispatch<MODE_CONV2D>(k, d, fast, s);
    if (d->mode == MODE_TEMPORAL) return dispatch<MODE_TEMPORAL>(k, d, fast, s);
    return dispatch<MODE_LINEAR>(k, d, fast, s);
}
