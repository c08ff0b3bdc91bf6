// Fused attention forward for head_dim 64 on gfx950 (fp16 in, fp32 softmax/accumulate, fp16 out).
//
// Replaces F.scaled_dot_product_attention at i2vgen-xl/pnp_utils.py:208-210 (spatial self-attention, and the
// cross-attention of the same processor class) and :314-316 (temporal self-attention), *including* the PnP
// query/key injection of :189-196 / :295-302: instead of copying the source branch's Q and K over the two target
// branches, batch element i simply reads Q/K of element i % qk_mod (aliasing; no HBM copy).
//
// One kernel serves all three attention flavours through strided row addressing (see anyv2v_hip.h): spatial
// tokens are contiguous, temporal sequences stride by H*W rows in the same channels-last matrix (no permute),
// cross-attention K/V are shared by the F frames of a clip (kv_div).
//
// Structure (per the MI355X guide's 32x32 flash ladder): 256 threads = 4 waves x 32 query rows; KV tiles of 64.
//   S^T = K Q^T   : v_mfma_f32_32x32x16_f16(a = K fragment, b = Q fragment): a lane owns ONE query column and 16
//                   of the 32 keys -> row max / row sum are in-lane plus one cross-half shuffle.
//   O^T = V^T P^T : a = V^T fragment (from a transposed LDS image), b = P fragment taken straight from the S^T
//                   registers (exp'ed, packed to fp16): no LDS round trip for P, per-lane O rescale.
// LDS: K tile [64 key][64 d], V^T tile [64 d][64 key-slot]; 16-byte chunks XOR-swizzled by (row>>1)&7 which is
// conflict-free for the 32-row ds_read_b128 fragment pattern.  Keys inside a V^T row are permuted (bits 2<->3)
// so that the 8 keys a lane needs for one MFMA K-step are one contiguous 16-byte read.
// Next KV tile is prefetched global->registers while the current one is consumed (async-stage split, T14).
#include <type_traits>

#include "common.h"

struct AttnK {
    const half_t* Q;
    const half_t* K;
    const half_t* V;
    half_t* O;
    int ldq, ldk, ldv, ldo;
    int batch, heads, Sq, Sk, inner;
    long long q_outer, q_inner, q_seq, kv_outer, kv_inner, kv_seq;
    int kv_div, qk_mod;
    float scale_log2;
    int q_tiles;
    int head_dim;
    int causal;  // generic kernel only: key j is visible to query s iff j <= s (CLIP text tower)
    int narrow_store;  // flash kernels: 8-byte epilogue stores (descriptor flag bit7; A/B of the widened epilogue)
};

__device__ __forceinline__ long long attn_row(long long i, int inner, long long so, long long si) {
    return (i / inner) * so + (i % inner) * si;
}

// ---------------------------------------------------------------------------------------------------------
// v2: same math / fragment layouts as flash_attn_d64_kernel, but K and V tiles are fetched by LDS-DMA
// (global_load_lds, 16 B per lane) into a 4-stage LDS ring three tiles ahead, with counted s_waitcnt vmcnt and ONE
// raw s_barrier per tile -- the v1 kernel's single register-prefetched tile exposed the global latency on every tile
// (~5000 cycles per 64-key tile for 512 cycles of MFMA).  Both tiles stay ROW-MAJOR in LDS ([64 key][64 d], 16-byte
// chunks XOR-swizzled on the DMA *source* side): K fragments are ds_read_b128 as before (swizzle (key>>1)&7), and
// the V^T fragments of O^T = V^T P^T come from ds_read_b64_tr_b16 (hardware 4x16 transpose, swizzle av_vswz(key)) -- no
// register transposition, no LDS stores at all in the main loop.
typedef __fp16 fp16x4v_t __attribute__((__vector_size__(8)));

__device__ __forceinline__ void glds16_attn(const half_t* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __attribute__((aligned(256))) half_t g_attn_zero_line[128];
// 16-byte-chunk swizzle of the row-major V tile.  ds_read_b64_tr_b16 is serviced in two 32-lane groups, each reading
// 4 consecutive keys x 64 bytes; keys r and r + 2 of such a block land on the same 32 banks (128-byte rows, 64 banks), so
// their chunk sets must be disjoint: bit 2 of the XOR mask = bit 1 of the key.  (`key & 7`, used before, gave a 2-way
// conflict on every transpose read: SQ_LDS_BANK_CONFLICT = 32 of 96 LDS cycles per KV tile, profiles/r01_attention_pmc.md.)
// Depends on key & 3 only, so the +8 / +16 t row offsets of the fragment reads stay immediates.
__device__ __forceinline__ int av_vswz(int key) { return (key & 3) | ((key & 2) << 1); }


// ds_read_b64_tr_b16 through inline asm: with the builtin hipcc treats the read as possibly aliasing the LDS-DMA
// writes still in flight and drains them with s_waitcnt vmcnt(0) every tile (seen in the .s), which defeats the
// prefetch ring.  An asm load is invisible to the compiler's waitcnt bookkeeping, so every consumer below sits
// behind an explicit s_waitcnt lgkmcnt(N) + sched_barrier (MI355X guide 5.7, form iii).
template <int OFF>
__device__ __forceinline__ fp16x4v_t lds_tr16(unsigned lds_addr) {
    fp16x4v_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "i"(OFF));
    return r;
}
__device__ __forceinline__ h8 join8(fp16x4v_t a, fp16x4v_t b) {
    h8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = (half_t)a[e];
        v[4 + e] = (half_t)b[e];
    }
    return v;
}

// NV = number of V / O streams that share one S = Q K^T and one softmax:
//   NV = 1 : plain attention (and PnP aliasing through qk_mod);
//   NV = 3 : the PnP injection step computed the way the reference's semantics allow: on
//            injection steps the uncond / cond branches use the SOURCE branch's Q and K (pnp_utils.py:189-196), so
//            P = softmax(Q_src K_src^T) is identical for all three branches -- one block computes it once per KV
//            tile and applies it to V_src, V_uncond, V_cond (1/3 of the QK^T MFMAs and 1/3 of the exp/VALU work,
//            which is what bounds this kernel at head_dim 64).  Exact: same products, same order per branch.
// NW = waves per block (4: 128 query rows; 8: 256 query rows sharing one K / V ring -- half the LDS-DMA and ring LDS per
//      query, 4 waves per SIMD at <= 128 VGPRs).
//
// Softmax form.  A KV tile is processed as FOUR 16-key online-softmax steps: the P V MFMAs of step t execute in the matrix
// pipe while the exp2 / add / convert VALU work of step t + 1 issues, and the V^T fragments are fetched 16 keys at a time.
// There is no per-step row maximum: softmax is shift-invariant and the maximum only guards the exponent range, so each step
// is exponentiated against the OLD running maximum, P = exp2(fma(s, c, -m c)), and the partial row sum that is needed anyway
// doubles as the range check -- all P >= 0, hence "sum <= 2^10" bounds every P of the lane (fp16 holds 2^16).  Only when some
// lane of the wave trips the check (and on the very first step, where no maximum is known) the wave takes the textbook path on
// the same score registers: true maximum, rescale of O and l, re-exponentiation.  Exact in the sense that P only differs by the
// rounding of a shifted exponent; on the UNet's data the branch is taken on the first step of a row and practically never again
// (tests/gpu_checks.py::check_attention_forced_rescale forces it).  Per score this leaves fma + exp2 + add + 1/2 cvt_pk
// (the textbook form had a max on top and a deferred-rescale test per tile): 154 -> 136 VALU per wave and KV tile (SQ_INSTS_VALU minus MFMA, profiles/r02_pmc_raw.txt).
// Measured forms that were dropped (profiles/r02_attn_softmax_forms_ab*.txt): Q pre-multiplied by scale * log2(e) with the
// running maximum subtracted by the MFMA (C operand = a 16-register block of -m, D != C) removes the fma as well and is as fast
// as this form, but costs a second fp16 rounding of Q (error 3e-4 -> 7e-4, growing with |s c|); one whole-tile softmax without
// the 16-key steps needs 185 VGPRs (2 waves per SIMD) or spills.
template <int STAGES, int NV, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? (NV == 1 ? 4 : 2) : (NV == 1 ? (STAGES == 2 ? 4 : 3) : 2)) void flash_attn_d64_v2_kernel(const AttnK p, const half_t* zeros) {
    constexpr int TILE_BYTES = 8192, STAGE_BYTES = (1 + NV) * TILE_BYTES, PRE = STAGES - 1;  // PRE tiles in flight
    constexpr int NT = 64 * NW, DI = 512 / NT;  // threads; LDS-DMA instructions per thread and 8-KiB tile matrix (512 chunks of 16 B)
    constexpr int LPT = DI * (1 + NV);          // LDS-DMA per thread per tile
    static_assert((PRE == 2 && (LPT == 4 || LPT == 2)) || PRE == 1, "vmcnt immediates below");
    __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int qt = bid % p.q_tiles;
    const int bh = bid / p.q_tiles;
    const int h = bh % p.heads;
    const int i = bh / p.heads;  // NV == 3: index of the source-branch element; branches are i + b * qk_mod

    const int iq = (NV == 1 && p.qk_mod > 0) ? i % p.qk_mod : i;
    const long long qbase = attn_row(iq, p.inner, p.q_outer, p.q_inner);
    const long long kbase = attn_row(iq / p.kv_div, p.inner, p.kv_outer, p.kv_inner);
    long long obase[NV], vbase[NV];
#pragma unroll
    for (int b = 0; b < NV; ++b) {
        const int ib = i + b * (NV == 1 ? 0 : p.qk_mod);
        obase[b] = attn_row(ib, p.inner, p.q_outer, p.q_inner);
        vbase[b] = attn_row(ib / p.kv_div, p.inner, p.kv_outer, p.kv_inner);
    }

    const int q0 = qt * (32 * NW) + w * 32;
    const bool wave_active = q0 < p.Sq;
    const int qrow = q0 + l31;
    h8 qf[4];
    {
        const int qr = qrow < p.Sq ? qrow : p.Sq - 1;
        const half_t* qp = p.Q + (qbase + (long long)qr * p.q_seq) * p.ldq + h * 64 + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const h8*)(qp + 16 * ks);
    }

    // DMA assignment: instruction t, chunk slot = t * NT + tid -> row = slot >> 3, physical chunk = tid & 7.
    // Wave-uniform tile pointers (SGPRs, advanced by scalar adds) + constant 32-bit per-lane byte offsets; only the
    // ragged last tile needs the per-row test (wave-uniform branch), every other issue is bare DMA instructions.
    const int drow = tid >> 3, dpc = tid & 7;
    const int ntiles = (p.Sk + 63) / 64;
    const char* kptr = (const char*)(p.K + kbase * p.ldk + h * 64);
    const char* vptr[NV];
#pragma unroll
    for (int b = 0; b < NV; ++b) vptr[b] = (const char*)(p.V + vbase[b] * p.ldv + h * 64);
    unsigned koff[DI], voff_g[DI];
#pragma unroll
    for (int t = 0; t < DI; ++t) {
        const int row = drow + (NT / 8) * t;
        koff[t] = (unsigned)(((long long)row * p.kv_seq * p.ldk + ((dpc ^ ((row >> 1) & 7)) << 3)) * 2);
        voff_g[t] = (unsigned)(((long long)row * p.kv_seq * p.ldv + ((dpc ^ av_vswz(row)) << 3)) * 2);
    }
    const long long kstep = 128ll * p.kv_seq * p.ldk, vstep = 128ll * p.kv_seq * p.ldv;  // bytes per 64-key tile
    const int w_s = __builtin_amdgcn_readfirstlane(w);  // wave index in an SGPR: the DMA's LDS base (m0) stays scalar
    int issue_key0 = 0;
    auto issue = [&](int stage) {
        char* st = smem + stage * STAGE_BYTES;
        if (issue_key0 + 64 <= p.Sk) {
#pragma unroll
            for (int t = 0; t < DI; ++t) {
                glds16_attn((const half_t*)(kptr + koff[t]), st + (t * NT + w_s * 64) * 16);
#pragma unroll
                for (int b = 0; b < NV; ++b)
                    glds16_attn((const half_t*)(vptr[b] + voff_g[t]), st + (1 + b) * TILE_BYTES + (t * NT + w_s * 64) * 16);
            }
        } else {
#pragma unroll
            for (int t = 0; t < DI; ++t) {
                const bool ok = issue_key0 + drow + (NT / 8) * t < p.Sk;  // only the last tile can be ragged
                glds16_attn(ok ? (const half_t*)(kptr + koff[t]) : zeros, st + (t * NT + w_s * 64) * 16);
#pragma unroll
                for (int b = 0; b < NV; ++b)
                    glds16_attn(ok ? (const half_t*)(vptr[b] + voff_g[t]) : zeros,
                                st + (1 + b) * TILE_BYTES + (t * NT + w_s * 64) * 16);
            }
        }
        kptr += kstep;
#pragma unroll
        for (int b = 0; b < NV; ++b) vptr[b] += vstep;
        issue_key0 += 64;
    };

    f16v oacc[NV][2];
#pragma unroll
    for (int b = 0; b < NV; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[b][0][r] = oacc[b][1][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;  // m_run is set by the very first step
    const float c = p.scale_log2;

    // V^T fragment addressing (ds_read_b64_tr_b16): lane i16 of a 16-lane group supplies &V[kb + (i16>>2)][dcol + 4(i16&3)]
    const int i16 = lane & 15;
    const int vrow = 4 * hi + (i16 >> 2);                          // + 8 (second read) + 16 t
    const int vfl = av_vswz(vrow);                                 // row swizzle, constant per lane
    const int vc0 = 2 * ((lane >> 4) & 1) + ((i16 & 3) >> 1);      // 16-byte chunk within the 32-d half
    int voff[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) voff[db] = vrow * 128 + (((4 * db + vc0) ^ vfl) << 4) + (i16 & 1) * 8;

    for (int t = 0; t < PRE && t < ntiles; ++t) issue(t);
    // Q fragments must be complete BEFORE the loop: otherwise hipcc places their vmcnt wait at the first MFMA inside
    // the loop, where it has to be vmcnt(0) and drains the DMA ring on every tile (seen in the .s)
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));
    int stage = 0;
    for (int j = 0; j < ntiles; ++j) {
        const int ahead = ntiles - 1 - j;  // tiles issued after tile j that may still be in flight (<= PRE - 1)
        if (PRE >= 2 && ahead >= 1 && LPT == 4)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // PRE == 2: one tile (LPT instructions) may stay in flight
        else if (PRE >= 2 && ahead >= 1)
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (j + PRE < ntiles) issue((stage + PRE) % STAGES);
        if (wave_active) {
            const char* Ks = smem + stage * STAGE_BYTES;
            const unsigned ks_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)Ks;
            // ---- S^T = K Q^T for this wave's 32 queries x 64 keys
            // K fragments as inline asm with counted lgkmcnt waits: the first block's four reads, then its MFMAs with the second
            // block's reads slotted in, each MFMA waiting only for its own fragment (hipcc would wait lgkmcnt(0) in front of every
            // MFMA while an LDS-DMA load is in flight -- see gemm.hip / tools/wait_probe.hip -- i.e. also for the read just issued)
            h8 kf[2][4];
            unsigned kaddr[2];
            int kfk[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int key = 32 * kb + l31;
                kaddr[kb] = ks_lds + key * 128;
                kfk[kb] = (key >> 1) & 7;
            }
            auto kread = [&](int kb, int ks) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(kf[kb][ks]) : "v"(kaddr[kb] + (((2 * ks + hi) ^ kfk[kb]) << 4)) : "memory");
            };
            auto kwait = [&](int n) {
                switch (n) {
                    case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
                }
            };
            f16v sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kread(0, ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kwait(3);  // reads issued behind fragment (0, ks): three
                __builtin_amdgcn_sched_barrier(0);
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][ks], qf[ks], sacc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                kread(1, ks);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kwait(3 - ks);
                __builtin_amdgcn_sched_barrier(0);
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1][ks], qf[ks], sacc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j == ntiles - 1 && (p.Sk & 63) != 0) {  // key tail: only the last tile can hold masked keys
                const int key_base = j * 64 + 4 * hi;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (key_base + 32 * kb + (r & 3) + 8 * (r >> 2) >= p.Sk) sacc[kb][r] = -1e30f;
            }
            // ---- four 16-key steps: score registers r0 .. r0 + 7 of block kb are keys 16 t + {0..3, 8..11} + 4 hi
            auto step = [&](auto tc) {
                constexpr int t = decltype(tc)::value, kb = t >> 1, r0 = 8 * (t & 1);
                // V^T fragments of this k-step for every branch and both d-halves (asm reads: invisible to hipcc's waitcnt
                // bookkeeping, consumed behind the explicit lgkmcnt(0) + sched_barrier below)
                fp16x4v_t vt[NV][2][2];
#pragma unroll
                for (int b = 0; b < NV; ++b) {
                    const unsigned vb_lds = ks_lds + (1 + b) * TILE_BYTES;
                    vt[b][0][0] = lds_tr16<2 * t * 1024>(vb_lds + voff[0]);
                    vt[b][0][1] = lds_tr16<(2 * t + 1) * 1024>(vb_lds + voff[0]);
                    vt[b][1][0] = lds_tr16<2 * t * 1024>(vb_lds + voff[1]);
                    vt[b][1][1] = lds_tr16<(2 * t + 1) * 1024>(vb_lds + voff[1]);
                }
                float e[8];
                float nmc = -m_run * c;
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) e[i8] = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r0 + i8], c, nmc));  // raw v_exp_f32
                float ps = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
                const bool first = j == 0 && t == 0;  // no maximum known yet
                // range check by the partial row sum (all P >= 0); !(x <= T) also catches inf / nan
                if (first || __any(!(ps <= 1024.0f))) {
                    float mx = sacc[kb][r0];
#pragma unroll
                    for (int i8 = 1; i8 < 8; ++i8) mx = fmaxf(mx, sacc[kb][r0 + i8]);
                    {  // the other 8 keys of the step sit in the partner half-wave: v_permlane32_swap (VALU, no LDS round trip)
                        const unsigned mu = __float_as_uint(mx);
                        const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
                        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                    }
                    if (first) {
                        m_run = mx;  // O = l = 0: nothing to rescale
                    } else {
                        const float m_new = fmaxf(m_run, mx);  // the maximum never moves down (alpha <= 1)
                        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
                        l_run *= alpha;
#pragma unroll
                        for (int b = 0; b < NV; ++b)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                oacc[b][0][r] *= alpha;
                                oacc[b][1][r] *= alpha;
                            }
                        m_run = m_new;
                    }
                    nmc = -m_run * c;
#pragma unroll
                    for (int i8 = 0; i8 < 8; ++i8) e[i8] = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r0 + i8], c, nmc));
                    ps = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
                }
                l_run += ps;
                h8 pf;  // already the B operand of O^T += V^T P^T
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) pf[i8] = (half_t)e[i8];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < NV; ++b) {
                    oacc[b][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(join8(vt[b][0][0], vt[b][0][1]), pf, oacc[b][0], 0, 0, 0);
                    oacc[b][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(join8(vt[b][1][0], vt[b][1][1]), pf, oacc[b][1], 0, 0, 0);
                }
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
        }
        if (++stage == STAGES) stage = 0;
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (wave_active && (p.ldo & 7) == 0 && !p.narrow_store) {
        // Row-per-lane epilogue, widened (MI355X guide T21): a lane holds channels 8 k + 4 hi .. + 3 of its query row for the eight
        // column groups k = 4 db + g, i.e. eight 8-byte pieces per row and half-wave.  One v_permlane32_swap per dword of a group
        // pair (k, k + 1) hands the lower half-wave channels 8 k .. 8 k + 7 and the upper one 8 k + 8 .. 8 k + 15: four 16-byte
        // stores per lane instead of eight 8-byte ones -- same bytes, same addresses, half the store instructions (the tail of a
        // block is store-ISSUE bound).  All 64 lanes take part in the swaps; rows past Sq are not stored.
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int b = 0; b < NV; ++b) {
            half_t* op = p.O + (obase[b] + (long long)qrow * p.q_seq) * p.ldo + h * 64 + 8 * hi;
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {   // group pair (2 kp, 2 kp + 1): db = kp >> 1, g = 2 (kp & 1), + 1
                unsigned w0[2], w1[2];
#pragma unroll
                for (int d2 = 0; d2 < 2; ++d2) {
                    h2 a, c;
                    a[0] = (half_t)(oacc[b][kp >> 1][8 * (kp & 1) + 2 * d2] * inv);
                    a[1] = (half_t)(oacc[b][kp >> 1][8 * (kp & 1) + 2 * d2 + 1] * inv);
                    c[0] = (half_t)(oacc[b][kp >> 1][8 * (kp & 1) + 4 + 2 * d2] * inv);
                    c[1] = (half_t)(oacc[b][kp >> 1][8 * (kp & 1) + 4 + 2 * d2 + 1] * inv);
                    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, c), false, false);
                    w0[d2] = r[0];
                    w1[d2] = r[1];
                }
                if (qrow < p.Sq) *(u4*)(op + 16 * kp) = (u4){w0[0], w0[1], w1[0], w1[1]};
            }
        }
    } else if (wave_active && qrow < p.Sq) {
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int b = 0; b < NV; ++b) {
            half_t* op = p.O + (obase[b] + (long long)qrow * p.q_seq) * p.ldo + h * 64 + 4 * hi;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)(oacc[b][db][4 * g + e] * inv);
                    *(h4*)(op + 32 * db + 8 * g) = o;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Short-sequence attention (S <= 16, head_dim 64): the temporal self-attention of TransformerTemporalModel at the
// benchmark's 16 frames (pnp_utils.py:247-334).  HBM-bound (4 x 2 KiB per (clip, pixel, head)); one WAVE owns one
// (batch element, head):
//   S^T[key][q] : 2 x v_mfma_f32_16x16x32_f16, K and Q fragments loaded straight from global (both are k-contiguous)
//   softmax     : lane (q = l&15, key group l>>4) holds 4 keys -> in-lane + 2 shuffles
//   O^T = V^T P^T : 4 x v_mfma_f32_16x16x16_f16; P^T is already the B operand; V^T fragments come from a row-major
//                 2 KiB LDS image of V through ds_read_b64_tr_b16 (hardware 4x16 transpose; semantics pinned by
//                 anyv2v_selftest): lane i of a 16-lane group supplies &V[4g + (i>>2)][d0 + 4(i&3)] and receives
//                 V[4g + 0..3][d0 + i].
typedef __fp16 fp16x4_t __attribute__((__vector_size__(8)));

// NB = 3: the PnP injection step (pnp_utils.py:295-302: the uncond / cond branches use the SOURCE branch's Q and K) as ONE wave per
// (source element, head): Q, K are read and the softmax is taken once, then the three branches' V are applied one after the other --
// the same products in the same order per branch as the aliasing form (bit-equal), 8 instead of 12 tensor passes over HBM.
template <int NB>
__global__ __launch_bounds__(256) void short_attn_d64_kernel(const AttnK p) {
    __shared__ __attribute__((aligned(16))) half_t vs[4][16 * 64];
    __shared__ __attribute__((aligned(16))) half_t os[4][16 * 72];  // output tile, rows padded to 144 bytes
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long unit = (long long)blockIdx.x * 4 + w;  // (batch element, head); NB = 3: (source element, head)
    if (unit >= (long long)(NB == 1 ? p.batch : p.qk_mod) * p.heads) return;      // wave-uniform; no block-level sync in this kernel
    const int h = (int)(unit % p.heads);
    const long long i = unit / p.heads;
    const long long iq = (NB == 1 && p.qk_mod > 0) ? i % p.qk_mod : i;
    const long long qbase = attn_row(iq, p.inner, p.q_outer, p.q_inner);
    const long long kbase = attn_row(iq / p.kv_div, p.inner, p.kv_outer, p.kv_inner);
    const int l15 = lane & 15, g = lane >> 4;
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // fragments: lane (row = l15, k-chunk g): 8 halves at column 8 g + 32 ks
    h8 qf[2], kf[2];
    {
        const int qr = l15 < p.Sq ? l15 : p.Sq - 1;
        const half_t* qp = p.Q + (qbase + (long long)qr * p.q_seq) * p.ldq + h * 64 + 8 * g;
        const half_t* kp = p.K + (kbase + (long long)l15 * p.kv_seq) * p.ldk + h * 64 + 8 * g;
        const bool kok = l15 < p.Sk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = *(const h8*)(qp + 32 * ks);
            kf[ks] = kok ? *(const h8*)(kp + 32 * ks) : zero8;
        }
    }
    // V rows of every branch: requested up front (8 lanes fetch one full 128-byte key row per instruction)
    h8 vv[NB][2];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const long long vbase = attn_row((i + b * (NB == 1 ? 0 : p.qk_mod)) / p.kv_div, p.inner, p.kv_outer, p.kv_inner);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int key = 8 * ps + (lane >> 3);
            vv[b][ps] = key < p.Sk ? *(const h8*)(p.V + (vbase + (long long)key * p.kv_seq) * p.ldv + h * 64 + (lane & 7) * 8) : zero8;
        }
    }
    // S^T = K Q^T: D[key = 4 g + r][q = l15]
    f4 s = {0.f, 0.f, 0.f, 0.f};
    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[0], qf[0], s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[1], qf[1], s, 0, 0, 0);
    const float c = p.scale_log2;
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (4 * g + r >= p.Sk) s[r] = -1e30f;
        mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    h4 pf;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float e = exp2f((s[r] - mx) * c);
        sum += e;
        pf[r] = (half_t)e;
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    half_t* const ow = os[w];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const long long obase = attn_row(i + b * (NB == 1 ? 0 : p.qk_mod), p.inner, p.q_outer, p.q_inner);
        // V -> LDS row-major [16 keys][64 d]; same-wave LDS traffic is ordered (in-order DS queue), the fences keep hipcc from
        // moving the accesses of one branch across those of the next
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) *(h8*)(&vs[w][(8 * ps + (lane >> 3)) * 64 + (lane & 7) * 8]) = vv[b][ps];
        // O^T[d][q] = sum_key V^T[d][key] P^T[key][q]
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // The MFMA leaves lane (q = l15, g) with 4 of every 16 output channels: stored directly that is 32-byte pieces of 16
        // different rows per instruction.  Turn the tile through LDS (row stride 144 bytes) so that 8 lanes write one
        // full 128-byte row segment: two 1-KiB store instructions instead of four scattered 512-byte ones.
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const half_t* src = &vs[w][(4 * g + (l15 >> 2)) * 64 + 16 * db + 4 * (l15 & 3)];
            const fp16x4_t vt = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)src);
            h4 vf;
#pragma unroll
            for (int j = 0; j < 4; ++j) vf[j] = (half_t)vt[j];
            f4 o = {0.f, 0.f, 0.f, 0.f};
            o = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf, o, 0, 0, 0);
            h4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[r] * inv);
            *(h4*)(ow + l15 * 72 + 16 * db + 4 * g) = ov;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int q = 8 * ps + (lane >> 3), ch = lane & 7;
            const h8 ov = *(const h8*)(ow + q * 72 + ch * 8);
            if (q < p.Sq) *(h8*)(p.O + (obase + (long long)q * p.q_seq) * p.ldo + h * 64 + ch * 8) = ov;
        }
        if (NB > 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next branch overwrites vs / os
    }
}

// ---------------------------------------------------------------------------------------------------------
// Generic reference kernel: one thread per (batch, head, query); any head_dim <= 160, any strides, optional causal mask, optional
// additive score bias [heads, Sq, Sk] (fp32, added to the scaled scores: SEINE's time_rel_pos_bias, seine/models/attention.py:887).
__global__ void attn_naive_kernel(const AttnK p, const float* __restrict__ bias) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)p.batch * p.heads * p.Sq;
    if (idx >= total) return;
    const int s = (int)(idx % p.Sq);
    const int h = (int)((idx / p.Sq) % p.heads);
    const int i = (int)(idx / ((long long)p.Sq * p.heads));
    const int D = p.head_dim;
    const int iq = p.qk_mod > 0 ? i % p.qk_mod : i;
    const long long qbase = attn_row(iq, p.inner, p.q_outer, p.q_inner);
    const long long obase = attn_row(i, p.inner, p.q_outer, p.q_inner);
    const long long kbase = attn_row(iq / p.kv_div, p.inner, p.kv_outer, p.kv_inner);
    const long long vbase = attn_row(i / p.kv_div, p.inner, p.kv_outer, p.kv_inner);
    float q[160], o[160];
    const half_t* qp = p.Q + (qbase + (long long)s * p.q_seq) * p.ldq + h * D;
    for (int d = 0; d < D; ++d) {
        q[d] = (float)qp[d];
        o[d] = 0.f;
    }
    float m = -1e30f, l = 0.f;
    const float c = p.scale_log2;
    const int kend = p.causal ? (s + 1 < p.Sk ? s + 1 : p.Sk) : p.Sk;
    const float* brow = bias ? bias + ((long long)h * p.Sq + s) * p.Sk : nullptr;
    for (int k = 0; k < kend; ++k) {
        const half_t* kp = p.K + (kbase + (long long)k * p.kv_seq) * p.ldk + h * D;
        float sc = 0.f;
        for (int d = 0; d < D; ++d) sc += q[d] * (float)kp[d];
        if (brow) sc += brow[k] * (1.4426950408889634f / c);   // the bias in units of the raw score (c = scale log2 e)
        const float mn = fmaxf(m, sc);
        const float a = exp2f((m - mn) * c);
        const float pv = exp2f((sc - mn) * c);
        const half_t* vp = p.V + (vbase + (long long)k * p.kv_seq) * p.ldv + h * D;
        for (int d = 0; d < D; ++d) o[d] = o[d] * a + pv * (float)vp[d];
        l = l * a + pv;
        m = mn;
    }
    half_t* op = p.O + (obase + (long long)s * p.q_seq) * p.ldo + h * D;
    const float inv = 1.0f / l;
    for (int d = 0; d < D; ++d) op[d] = (half_t)(o[d] * inv);
}

// ---------------------------------------------------------------------------------------------------------
// Whole-sequence MFMA attention for SHORT sequences and any head_dim <= 128 that is a multiple of 16 (the CLIP towers of
// encode_prompt / _encode_image, pipeline_i2vgen_xl.py:224-441: text 16 heads x 64, 77 tokens, causal; vision 16 heads x 80,
// 257 tokens): the one-thread-per-query kernel above spent 3.8 ms per ViT-H layer (145 ms for a clip's two towers).
// One block = 4 waves = 64 queries of one (batch element, head); K ([key][dpad], zero-padded) and V^T ([d][key]) of that head
// sit in LDS, so a wave holds the complete score row block S^T = K Q^T of its 16 queries in registers (NKB 16-key blocks),
// takes the exact softmax over it (no online rescaling), and multiplies O^T = V^T P^T.  The P registers feed the second MFMA
// directly: a lane's scores are keys {4 lq + r + 16 kb}, and the contraction index of an MFMA may be any permutation as long as
// both operands use it -- the V^T fragment is read in that key order (two 8-byte LDS reads).
template <int DS, int NKB, bool BIAS = false>   // DS = dpad / 32 (2..5), NKB = padded keys / 16 (even: PV contracts 32 keys per MFMA)
__global__ __launch_bounds__(256) void small_attn_mfma_kernel(const AttnK p, const float* __restrict__ bias = nullptr) {
    constexpr int DPAD = DS * 32, SKP = NKB * 16;
    constexpr int KLD = DPAD + 8;    // halves per K row (+16 B: breaks the power-of-2 row stride for the ds_read_b128 fragments)
    constexpr int VLD = SKP + 4;     // halves per V^T row
    extern __shared__ __attribute__((aligned(16))) char smem_sa[];
    half_t* const Ks = (half_t*)smem_sa;                  // [SKP][KLD]
    half_t* const Vt = Ks + SKP * KLD;                    // [DPAD][VLD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int qt = blockIdx.x, h = blockIdx.y, i = blockIdx.z;
    const int D = p.head_dim;
    const int iq = p.qk_mod > 0 ? i % p.qk_mod : i;
    const long long qbase = attn_row(iq, p.inner, p.q_outer, p.q_inner);
    const long long obase = attn_row(i, p.inner, p.q_outer, p.q_inner);
    const long long kbase = attn_row(iq / p.kv_div, p.inner, p.kv_outer, p.kv_inner);
    const long long vbase = attn_row(i / p.kv_div, p.inner, p.kv_outer, p.kv_inner);
    // ---- K -> LDS row-major (zero rows / columns beyond Sk / D), V -> LDS transposed
    const int dch = DPAD / 8;   // 16-byte chunks per padded row
    for (int c = tid; c < SKP * dch; c += 256) {
        const int key = c / dch, d0 = (c - key * dch) * 8;
        h8 kv = (h8){0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
        if (key < p.Sk && d0 < D) {   // (D % 8 == 0: a chunk is entirely inside or outside the head)
            kv = *(const h8*)(p.K + (kbase + (long long)key * p.kv_seq) * p.ldk + h * D + d0);
            vv = *(const h8*)(p.V + (vbase + (long long)key * p.kv_seq) * p.ldv + h * D + d0);
        }
        *(h8*)(Ks + key * KLD + d0) = kv;
#pragma unroll
        for (int e = 0; e < 8; ++e) Vt[(d0 + e) * VLD + key] = vv[e];
    }
    __syncthreads();
    // ---- S^T = K Q^T for this wave's 16 queries
    const int q0 = qt * 64 + w * 16;
    const int qrow = q0 + l15;
    h8 qf[DS];
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) {
        const int d0 = ds * 32 + lq * 8;
        qf[ds] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
        if (qrow < p.Sq && d0 < D) qf[ds] = *(const h8*)(p.Q + (qbase + (long long)qrow * p.q_seq) * p.ldq + h * D + d0);
    }
    f4 sc[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        sc[kb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
            const h8 kf = *(const h8*)(Ks + (kb * 16 + l15) * KLD + ds * 32 + lq * 8);
            sc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ds], sc[kb], 0, 0, 0);
        }
    }
    // lane: query l15, keys 16 kb + 4 lq + r.  Mask (padding, causal), exact softmax over the whole row.
    float m = -1e30f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kb * 16 + 4 * lq + r;
            const bool ok = key < p.Sk && (!p.causal || key <= qrow);
            if (BIAS && ok && qrow < p.Sq)   // the bias in units of the raw score (logit = scale * s + bias; c = scale * log2 e)
                sc[kb][r] += bias[((long long)h * p.Sq + qrow) * p.Sk + key] * (1.4426950408889634f / p.scale_log2);
            sc[kb][r] = ok ? sc[kb][r] : -1e30f;
            m = fmaxf(m, sc[kb][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
    const float c = p.scale_log2;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = sc[kb][r] > -1e29f ? exp2f((sc[kb][r] - m) * c) : 0.f;
            sc[kb][r] = e;
            l += e;
        }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // ---- O^T = V^T P^T: 32 keys per MFMA in the order (4 lq + r) | (16 + 4 lq + r) of the key-block pair
    f4 o[DPAD / 16];
#pragma unroll
    for (int db = 0; db < DPAD / 16; ++db) o[db] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < NKB / 2; ++kp) {
        h8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pf[r] = (half_t)sc[2 * kp][r];
            pf[4 + r] = (half_t)sc[2 * kp + 1][r];
        }
#pragma unroll
        for (int db = 0; db < DPAD / 16; ++db) {
            const half_t* vr = Vt + (db * 16 + l15) * VLD + kp * 32 + 4 * lq;
            const h4 v0 = *(const h4*)vr, v1 = *(const h4*)(vr + 16);
            const h8 vf = (h8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[db], 0, 0, 0);
        }
    }
    // lane: query l15, output dims 16 db + 4 lq + r
    if (qrow < p.Sq) {
        half_t* op = p.O + (obase + (long long)qrow * p.q_seq) * p.ldo + h * D;
#pragma unroll
        for (int db = 0; db < DPAD / 16; ++db) {
            const int d0 = db * 16 + 4 * lq;
            if (d0 < D) {
                h4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[db][r] * inv);
                *(h4*)(op + d0) = ov;
            }
        }
    }
}

static int fill(const AnyV2VAttnDesc* d, AttnK& k, int head_dim) {
    AV_CHECK(d != nullptr, "attention: null descriptor");
    AV_CHECK(d->Q && d->K && d->V && d->O, "attention: null pointer");
    AV_CHECK(d->batch > 0 && d->heads > 0 && d->Sq > 0 && d->Sk > 0, "attention: bad sizes");
    AV_CHECK(d->inner > 0 && d->kv_div > 0 && d->qk_mod >= 0, "attention: bad inner/kv_div/qk_mod");
    k.Q = (const half_t*)d->Q;
    k.K = (const half_t*)d->K;
    k.V = (const half_t*)d->V;
    k.O = (half_t*)d->O;
    k.ldq = d->ldq; k.ldk = d->ldk; k.ldv = d->ldv; k.ldo = d->ldo;
    k.batch = d->batch; k.heads = d->heads; k.Sq = d->Sq; k.Sk = d->Sk; k.inner = d->inner;
    k.q_outer = d->q_outer; k.q_inner = d->q_inner; k.q_seq = d->q_seq;
    k.kv_outer = d->kv_outer; k.kv_inner = d->kv_inner; k.kv_seq = d->kv_seq;
    k.kv_div = d->kv_div; k.qk_mod = d->qk_mod;
    k.scale_log2 = d->scale * 1.4426950408889634f;
    k.q_tiles = (d->Sq + 127) / 128;
    k.head_dim = head_dim;
    k.causal = 0;
    k.narrow_store = (d->flags & 128) ? 1 : 0;
    return ANYV2V_OK;
}

static int launch_naive(const AttnK& k, hipStream_t s, const float* bias = nullptr) {
    const long long total = (long long)k.batch * k.heads * k.Sq;
    hipLaunchKernelGGL(attn_naive_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, s, k, bias);
    return av_launch_status("attention_naive");
}

// PnP injection launch (i2vgen-xl/pnp_utils.py:189-196, :295-302): three branches whose Q / K are the source branch's -> the shared-softmax
// kernels.  ONE predicate for the short and the flash path (flag bit3 forces the per-branch aliasing form).
static inline bool av_attn_pnp3(const AttnK& k, const AnyV2VAttnDesc* d) {
    return k.qk_mod > 0 && k.batch == 3 * k.qk_mod && k.kv_div == 1 && !(d->flags & 8);
}

extern "C" int anyv2v_attention_f16(const AnyV2VAttnDesc* d, void* stream) {
    AttnK k;
    int rc = fill(d, k, 64);
    if (rc != ANYV2V_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const bool fast = !(d->flags & 1) && d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 4 == 0 &&
                      av_aligned16(d->Q) && av_aligned16(d->K) && av_aligned16(d->V) && av_aligned16(d->O);
    if (!fast) return launch_naive(k, s);
    if (k.Sq <= 16 && k.Sk <= 16 && d->ldo % 8 == 0 && !(d->flags & 2)) {  // temporal attention at <= 16 frames: one wave per sequence
        if (av_attn_pnp3(k, d)) {
            // PnP injection step: one wave per (source element, head), one softmax, three V / O streams (flag bit3: aliasing form)
            const long long units3 = (long long)k.qk_mod * k.heads;
            hipLaunchKernelGGL(short_attn_d64_kernel<3>, dim3((unsigned)((units3 + 3) / 4)), dim3(256), 0, s, k);
            return av_launch_status("short_attn_d64<pnp3>");
        }
        const long long units = (long long)k.batch * k.heads;
        hipLaunchKernelGGL(short_attn_d64_kernel<1>, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, k);
        return av_launch_status("short_attn_d64");
    }
    const long long nwg = (long long)k.batch * k.heads * k.q_tiles;
    AV_CHECK(nwg < (1ll << 31), "attention: grid too large");
    static const half_t* zeros = nullptr;
    if (zeros == nullptr) {
        void* ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_attn_zero_line)) == hipSuccess) zeros = (const half_t*)ptr;
    }
    AV_CHECK(zeros != nullptr, "attention: zero line symbol unavailable");
    if (av_attn_pnp3(k, d)) {
        // PnP injection step: one softmax per source element, three V / O streams (flag bit3 forces the aliasing form)
        const long long nwg3 = (long long)k.qk_mod * k.heads * k.q_tiles;
        // long KV loops with enough 256-query blocks to fill the chip (the 64x64 level: 16 x 5 x 16 = 1280): 8-wave blocks around a
        // 3-stage ring -- two K / V tile sets in flight instead of one, half the LDS-DMA per query, one 96-KiB block per CU (the same
        // two waves per SIMD as two 4-wave blocks).  0.6845 -> 0.6479 ms at (48, 5, 4096); 7 % slower at (48, 10, 1024), which stays
        // on the 4-wave form (profiles/r05_attn_pnp_8wave_ab.txt); bit-equal.  Flag bit2: never (as for the plain kernel).
        const long long n8 = (long long)k.qk_mod * k.heads * ((d->Sq + 255) / 256);
        if (!(d->flags & 4) && d->Sk >= 2048 && d->Sq >= 2048 && n8 >= 1024) {
            k.q_tiles = (d->Sq + 255) / 256;
            hipLaunchKernelGGL((flash_attn_d64_v2_kernel<3, 3, 8>), dim3((unsigned)n8), dim3(512), 0, s, k, zeros);
            return av_launch_status("flash_attn_d64_v2<pnp3, 8 waves>");
        }
        hipLaunchKernelGGL((flash_attn_d64_v2_kernel<2, 3, 4>), dim3((unsigned)nwg3), dim3(256), 0, s, k, zeros);
        return av_launch_status("flash_attn_d64_v2<pnp3>");
    }
    // 8-wave blocks (256 query rows per K / V ring) for launches that still fill the chip with them and whose KV loop is long
    // enough to amortise the bigger block (measured: 1.19 -> 1.09 ms at 48 x 5 x 4096^2, equal at 1024^2, slower at Sk = 145)
    const long long nwg8 = (long long)k.batch * k.heads * ((d->Sq + 255) / 256);
    // (round 5: from Sk = 2048 on -- at 48 x 10 x 1024^2 the 4-wave kernel on its 2-stage ring is 4 % faster, 146.4 vs 152.0 us)
    if (!(d->flags & 4) && d->Sk >= 2048 && d->Sq >= 1024 && nwg8 >= 1024) {
        k.q_tiles = (d->Sq + 255) / 256;
        hipLaunchKernelGGL((flash_attn_d64_v2_kernel<3, 1, 8>), dim3((unsigned)nwg8), dim3(512), 0, s, k, zeros);
        return av_launch_status("flash_attn_d64_v2<8 waves>");
    }
    // 4-wave blocks on a 2-stage ring: 32 KiB of LDS per block = four blocks per CU (three with 3 stages; the kernel's launch bounds ask
    // for four waves per SIMD, it allocates 126 VGPRs -- profiles/r06_attention_resource_usage.txt); the short-KV launches
    // (cross-attention, Sk = 145: three tiles per block, 7680 blocks) are bound by how many blocks are in flight -- 95.8 -> 82.6 us at
    // (48, 5, 4096, 145), never slower up to Sk = 1024 (profiles/r05_attn_cross_2stage_ab.txt); bit-equal
    hipLaunchKernelGGL((flash_attn_d64_v2_kernel<2, 1, 4>), dim3((unsigned)nwg), dim3(256), 0, s, k, zeros);
    return av_launch_status("flash_attn_d64_v2");
}

// Any sequence length at head_dim <= 160 (a multiple of 8): the whole-sequence kernel above with a loop over 96-key blocks and the
// online softmax (running maximum / sum per query, accumulators rescaled when the maximum moves).  SEINE's spatial attention runs
// here: 8 heads of 40 / 80 / 160 channels over 2560 / 640 tokens (the head_dim-64 flash kernel does not apply, and the
// one-thread-per-query kernel took 97 % of that model's forward).  Not tuned: K / V of a block are staged through registers, V is
// transposed by scalar LDS stores.
template <int DS>
__global__ __launch_bounds__(256) void loop_attn_mfma_kernel(const AttnK p) {
    constexpr int NKB = 6, DPAD = DS * 32, SKP = NKB * 16;
    constexpr int KLD = DPAD + 8, VLD = SKP + 4;
    extern __shared__ __attribute__((aligned(16))) char smem_la[];
    half_t* const Ks = (half_t*)smem_la;                  // [SKP][KLD]
    half_t* const Vt = Ks + SKP * KLD;                    // [DPAD][VLD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int qt = blockIdx.x, h = blockIdx.y, i = blockIdx.z;
    const int D = p.head_dim;
    const int iq = p.qk_mod > 0 ? i % p.qk_mod : i;
    const long long qbase = attn_row(iq, p.inner, p.q_outer, p.q_inner);
    const long long obase = attn_row(i, p.inner, p.q_outer, p.q_inner);
    const long long kbase = attn_row(iq / p.kv_div, p.inner, p.kv_outer, p.kv_inner);
    const long long vbase = attn_row(i / p.kv_div, p.inner, p.kv_outer, p.kv_inner);
    const int q0 = qt * 64 + w * 16;
    const int qrow = q0 + l15;
    h8 qf[DS];
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) {
        const int d0 = ds * 32 + lq * 8;
        qf[ds] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
        if (qrow < p.Sq && d0 < D) qf[ds] = *(const h8*)(p.Q + (qbase + (long long)qrow * p.q_seq) * p.ldq + h * D + d0);
    }
    f4 o[DPAD / 16];
#pragma unroll
    for (int db = 0; db < DPAD / 16; ++db) o[db] = (f4){0.f, 0.f, 0.f, 0.f};
    float m = -1e30f, l = 0.f;
    const float c = p.scale_log2;
    const int dch = DPAD / 8;
    // causal: blocks entirely above the diagonal of this query block hold nothing visible
    const int k_end = p.causal ? min(p.Sk, qt * 64 + 64) : p.Sk;
    for (int k0 = 0; k0 < k_end; k0 += SKP) {
        __syncthreads();   // the previous block's fragments are read
        for (int cidx = tid; cidx < SKP * dch; cidx += 256) {
            const int key = cidx / dch, d0 = (cidx - key * dch) * 8;
            h8 kv = (h8){0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
            if (k0 + key < p.Sk && d0 < D) {
                kv = *(const h8*)(p.K + (kbase + (long long)(k0 + key) * p.kv_seq) * p.ldk + h * D + d0);
                vv = *(const h8*)(p.V + (vbase + (long long)(k0 + key) * p.kv_seq) * p.ldv + h * D + d0);
            }
            *(h8*)(Ks + key * KLD + d0) = kv;
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[(d0 + e) * VLD + key] = vv[e];
        }
        __syncthreads();
        f4 sc[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            sc[kb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) {
                const h8 kf = *(const h8*)(Ks + (kb * 16 + l15) * KLD + ds * 32 + lq * 8);
                sc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ds], sc[kb], 0, 0, 0);
            }
        }
        float mb = -1e30f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + kb * 16 + 4 * lq + r;
                const bool ok = key < p.Sk && (!p.causal || key <= qrow);
                sc[kb][r] = ok ? sc[kb][r] : -1e30f;
                mb = fmaxf(mb, sc[kb][r]);
            }
        mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float m_new = fmaxf(m, mb);
        const float alpha = exp2f((m - m_new) * c);     // 1 when the maximum stays; 0 for the first block (m = -1e30)
        m = m_new;
        float lb = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = sc[kb][r] > -1e29f ? exp2f((sc[kb][r] - m) * c) : 0.f;
                sc[kb][r] = e;
                lb += e;
            }
        lb += __shfl_xor(lb, 16, 64);
        lb += __shfl_xor(lb, 32, 64);
        l = l * alpha + lb;
#pragma unroll
        for (int db = 0; db < DPAD / 16; ++db)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[db][r] *= alpha;
#pragma unroll
        for (int kp = 0; kp < NKB / 2; ++kp) {
            h8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (half_t)sc[2 * kp][r];
                pf[4 + r] = (half_t)sc[2 * kp + 1][r];
            }
#pragma unroll
            for (int db = 0; db < DPAD / 16; ++db) {
                const half_t* vr = Vt + (db * 16 + l15) * VLD + kp * 32 + 4 * lq;
                const h4 v0 = *(const h4*)vr, v1 = *(const h4*)(vr + 16);
                const h8 vf = (h8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[db], 0, 0, 0);
            }
        }
    }
    if (qrow < p.Sq) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        half_t* op = p.O + (obase + (long long)qrow * p.q_seq) * p.ldo + h * D;
#pragma unroll
        for (int db = 0; db < DPAD / 16; ++db) {
            const int d0 = db * 16 + 4 * lq;
            if (d0 < D) {
                h4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[db][r] * inv);
                *(h4*)(op + d0) = ov;
            }
        }
    }
}

extern "C" int anyv2v_attention_small_f16(const AnyV2VAttnDesc* d, int32_t head_dim, void* stream) {
    AV_CHECK(head_dim > 0 && head_dim <= 160, "attention_small: head_dim must be in 1..160");
    AttnK k;
    int rc = fill(d, k, head_dim);
    if (rc != ANYV2V_OK) return rc;
    k.causal = (d->flags & 16) ? 1 : 0;
    // short sequences, head_dim a multiple of 8 (16-byte chunks of a head are loaded whole): whole-sequence MFMA kernel (K and V^T of
    // a head in LDS, zero-padded to 32 DS columns); flag bit0 = naive kernel.  DS = 5 (head_dim 136..160, ConsistI2V's temporal
    // attention at C = 1280 / 8 heads) fits the LDS with up to 96 keys only.
    const int ds = (head_dim + 31) / 32;
    const int nkb = d->Sk <= 96 ? 6 : (d->Sk <= 288 && ds <= 4 ? 18 : 0);
    const bool fast = !(d->flags & 1) && head_dim % 8 == 0 && ds >= 2 && ds <= 5 && nkb > 0 && d->ldq % 8 == 0 && d->ldk % 8 == 0 &&
                      d->ldv % 8 == 0 && d->ldo % 4 == 0 && av_aligned16(d->Q) && av_aligned16(d->K) && av_aligned16(d->V) &&
                      (((uintptr_t)d->O) & 7) == 0 && d->heads <= 65535 && d->batch <= 65535;
    // the same conditions with a longer key sequence: the 96-key loop kernel (online softmax)
    const bool loop_ok = !(d->flags & 1) && head_dim % 8 == 0 && ds >= 1 && ds <= 5 && nkb == 0 && d->ldq % 8 == 0 && d->ldk % 8 == 0 &&
                         d->ldv % 8 == 0 && d->ldo % 4 == 0 && av_aligned16(d->Q) && av_aligned16(d->K) && av_aligned16(d->V) &&
                         (((uintptr_t)d->O) & 7) == 0 && d->heads <= 65535 && d->batch <= 65535;
    if (loop_ok) {
        const dim3 lgrid((unsigned)((d->Sq + 63) / 64), (unsigned)d->heads, (unsigned)d->batch);
        const int dsl = ds < 2 ? 2 : ds;
        const size_t llds = (size_t)96 * (dsl * 32 + 8) * 2 + (size_t)(dsl * 32) * (96 + 4) * 2;
#define AV_LA(DS_)                                                                                                            \
    do {                                                                                                                      \
        static bool attr_set = false;                                                                                         \
        if (!attr_set) {                                                                                                      \
            (void)hipFuncSetAttribute((const void*)loop_attn_mfma_kernel<DS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds); \
            attr_set = true;                                                                                                  \
        }                                                                                                                     \
        hipLaunchKernelGGL((loop_attn_mfma_kernel<DS_>), lgrid, dim3(256), llds, (hipStream_t)stream, k);                     \
    } while (0)
        if (dsl == 2) AV_LA(2); else if (dsl == 3) AV_LA(3); else if (dsl == 4) AV_LA(4); else AV_LA(5);
#undef AV_LA
        return av_launch_status("loop_attn_mfma");
    }
    if (!fast) return launch_naive(k, (hipStream_t)stream);
    const dim3 grid((unsigned)((d->Sq + 63) / 64), (unsigned)d->heads, (unsigned)d->batch);
    const size_t lds = (size_t)(nkb * 16) * (ds * 32 + 8) * 2 + (size_t)(ds * 32) * (nkb * 16 + 4) * 2;
#define AV_SA(DS_, NKB_)                                                                                                   \
    do {                                                                                                                   \
        static bool attr_set = false;                                                                                      \
        if (!attr_set) {                                                                                                   \
            (void)hipFuncSetAttribute((const void*)small_attn_mfma_kernel<DS_, NKB_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                                 \
            attr_set = true;                                                                                               \
        }                                                                                                                  \
        hipLaunchKernelGGL((small_attn_mfma_kernel<DS_, NKB_>), grid, dim3(256), lds, (hipStream_t)stream, k);            \
    } while (0)
    if (nkb == 6) {
        if (ds == 2) AV_SA(2, 6); else if (ds == 3) AV_SA(3, 6); else if (ds == 4) AV_SA(4, 6); else AV_SA(5, 6);
    } else {
        if (ds == 2) AV_SA(2, 18); else if (ds == 3) AV_SA(3, 18); else AV_SA(4, 18);
    }
#undef AV_SA
    return av_launch_status("small_attn_mfma");
}

// Attention with an additive score bias (fp32 [heads, Sq, Sk], added to scale * q.k before the softmax): the generic kernel.
extern "C" int anyv2v_attention_bias_f16(const AnyV2VAttnDesc* d, int32_t head_dim, const float* bias, void* stream) {
    AV_CHECK(head_dim > 0 && head_dim <= 160, "attention_bias: head_dim must be in 1..160");
    AV_CHECK(bias != nullptr, "attention_bias: null bias");
    AV_CHECK(d && d->scale > 0.f, "attention_bias: scale must be positive");
    {   // short sequences at head_dim % 8 == 0: the whole-sequence MFMA kernel with the bias added to the score fragments
        const int ds = (head_dim + 31) / 32;
        if (!(d->flags & 1) && head_dim % 8 == 0 && ds >= 2 && ds <= 5 && d->Sk <= 96 && d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 &&
            d->ldo % 4 == 0 && av_aligned16(d->Q) && av_aligned16(d->K) && av_aligned16(d->V) && (((uintptr_t)d->O) & 7) == 0 &&
            d->heads <= 65535 && d->batch <= 65535 && !(d->flags & 16)) {
            AttnK kb;
            int rcb = fill(d, kb, head_dim);
            if (rcb != ANYV2V_OK) return rcb;
            kb.causal = 0;
            const dim3 grid((unsigned)((d->Sq + 63) / 64), (unsigned)d->heads, (unsigned)d->batch);
            const size_t lds = (size_t)96 * (ds * 32 + 8) * 2 + (size_t)(ds * 32) * (96 + 4) * 2;
#define AV_SB(DS_)                                                                                                                       \
    do {                                                                                                                                 \
        static bool attr_set = false;                                                                                                    \
        if (!attr_set) {                                                                                                                 \
            (void)hipFuncSetAttribute((const void*)small_attn_mfma_kernel<DS_, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                                             \
        }                                                                                                                                \
        hipLaunchKernelGGL((small_attn_mfma_kernel<DS_, 6, true>), grid, dim3(256), lds, (hipStream_t)stream, kb, bias);                \
    } while (0)
            if (ds == 2) AV_SB(2); else if (ds == 3) AV_SB(3); else if (ds == 4) AV_SB(4); else AV_SB(5);
#undef AV_SB
            return av_launch_status("small_attn_mfma<bias>");
        }
    }
    AttnK k;
    int rc = fill(d, k, head_dim);
    if (rc != ANYV2V_OK) return rc;
    k.causal = (d->flags & 16) ? 1 : 0;
    return launch_naive(k, (hipStream_t)stream, bias);
}
