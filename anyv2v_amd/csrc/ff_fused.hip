// Fused feed-forward of the 320-channel transformer blocks (gfx950): y = GEGLU(x W1^T + b1) W2^T + b2 (+ residual) in ONE
// kernel -- the [tokens, 1280] hidden activation (503 MB per launch at 16 f x 512^2, B = 3) is never written to or read from HBM.
//
// Replaces (reference = TIGER-AI-Lab/AnyV2V): diffusers-0.26.3 FeedForward (GEGLU -> Linear) of BasicTransformerBlock behind
// i2vgen-xl/pipelines/pipeline_i2vgen_xl.py:1146; in-tree restatement consisti2v/consisti2v/models/videoldm_transformer_blocks.py:
// 545-563 (norm3 -> ff -> + residual).  Unfused this is gemm_ws_kernel<GEGLU> (373 us) + gemm_big_kernel (206 us) per launch pair.
//
// Structure.  The hidden dimension is walked in 40 SLABS of 32 units.  A block (8 waves) owns four 32-row strips; a strip belongs to a
// PAIR of waves (w, w ^ 1) that share it as follows:
//   phase A(j): hidden slab j = GEGLU(x W1_j^T): 32 rows x [16 h | 16 gate] columns per wave (wave a: hidden 0..15 of the slab, wave b:
//               16..31); the strip's activations x[32][320] stay in REGISTERS for the whole strip (80 VGPRs of MFMA b-fragments);
//   exchange  : the 16 hidden values a wave does not own come from its partner through 1 KB of LDS per wave and step;
//   phase B(j): y[32 rows][160 columns per wave] += hidden_j W2_j^T (80 accumulator VGPRs, live for the whole strip).
// The GEGLU output registers of phase A ARE the MFMA b-fragment of phase B: a lane holds hidden units 4 lq + r of token l15, and the
// K-slot order inside an MFMA is free as long as both operands agree, so W2 is packed on the host with its 32 hidden columns per slab
// permuted to (slot 8 lq + e -> hidden 4 lq + e for e < 4, 16 + 4 lq + e - 4 for e >= 4) -- no LDS turn, no cross-lane move.
// Weights are STREAMED: W1_j (64 x 320, 40 KB) and W2_j (320 x 32, 20 KB) arrive by LDS-DMA into two-deep rings, one slab step
// ahead, the same stream for every block (2.4 MB, L2-resident); one s_barrier per slab step.  Phase B runs one step behind phase A
// (B(j-1) next to A(j)), and the two SIMD partners (waves w, w + 4) run the two phases in opposite order, so one wave's GEGLU VALU
// work sits beside the other's MFMAs.
// Numerics: fp32 accumulation, hidden rounded to fp16 once (as the unfused GEGLU kernel stores it), y = fp16(acc + b2) then + residual
// in fp16 -- the unfused pair's rounding points; only the summation order inside a 32-wide K-step differs.
#include "gemm_common.h"

namespace {
constexpr int FF_C = 320, FF_H = 1280, FF_HS = 32;
constexpr int FF_NSLAB = FF_H / FF_HS;               // 40
constexpr int FF_W1_BYTES = 2 * FF_HS * FF_C * 2;    // 64 rows x 640 B
constexpr int FF_W2_BYTES = FF_C * FF_HS * 2;        // 320 rows x 64 B
constexpr int FF_X_BYTES = 8 * 1024;                 // exchange: 8 waves x 2 row fragments x 64 lanes x 8 B
constexpr int FF_B1_BYTES = 2 * FF_H * 2, FF_B2_BYTES = FF_C * 2;
constexpr int FF_LDS = 2 * FF_W1_BYTES + 2 * FF_W2_BYTES + 2 * FF_X_BYTES + FF_B1_BYTES + FF_B2_BYTES;

struct FFK {
    const half_t* X;
    const half_t* W1;   // [2560][320], rows interleaved [16 h | 16 gate] (GEGLU.pack)
    const half_t* b1;   // [2560], same interleave
    const half_t* W2;   // [40][320][32]: slab-major, hidden columns of a slab in MFMA slot order (see above)
    const half_t* b2;   // [320]
    const half_t* R;    // residual [M][ldr] or nullptr
    half_t* Y;
    int M, ldx, ldr, ldy, nstrips;
};

__device__ __forceinline__ h8 ffr128(unsigned addr, int off) {   // off: a constant after unrolling (16-bit immediate)
    h8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off) : "memory");
    return v;
}
__device__ __forceinline__ h4 ffr64(unsigned addr) {
    h4 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void ffw64(unsigned addr, h4 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
#ifdef ANYV2V_EXPERIMENTS
__device__ long long g_ff_trace[2 * 8 * 16];   // [wave 0 | wave 4 of block 0][step 4..11 of round 1][stamp]
#endif
#define FF_LGKM(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
}  // namespace

// KO (probe build only, tools/ff_fused_ab.py): 1 = no LDS-DMA (stale weights), 2 = GEGLU replaced by h * gate, 3 = no phase-B MFMAs,
// 4 = no phase-A MFMAs, 5 = no step barrier (races; timing only), 6 = no exchange, 7 = s_memtime stamps per phase.
// VAR: bit0 = fragment rings one step deeper (measured: no gain -- the step is bound by instruction ISSUE of the two SIMD partners
// together, not by LDS latency), bit1 = s_setprio 1 around the MFMA loops (-2.5 %, default).  profiles/r04_ff_fused_ab_v1.txt
template <bool RES, int KO = 0, int VAR = 2>
__global__ __launch_bounds__(512) void ff_fused_c320_kernel(const FFK p) {
    __shared__ __attribute__((aligned(16))) char smem[FF_LDS];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    constexpr unsigned W1B = 0, W2B = 2 * FF_W1_BYTES, XB = W2B + 2 * FF_W2_BYTES, B1B = XB + 2 * FF_X_BYTES, B2B = B1B + FF_B1_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int half_ = w & 1, pr = w >> 1;            // position in the pair, pair of the block
    const int late = w >> 2;                         // waves 4-7 (SIMD partners of 0-3): their step barrier sits between the phases
    const int G = gridDim.x;

    // ---- biases into LDS once (plain loads: no LDS-DMA is in flight yet)
    for (int i = tid; i < (FF_B1_BYTES + FF_B2_BYTES) / 16; i += 512) {
        const h8 v = i < FF_B1_BYTES / 16 ? *(const h8*)(p.b1 + i * 8) : *(const h8*)(p.b2 + (i - FF_B1_BYTES / 16) * 8);
        *(h8*)(smem + B1B + i * 16) = v;
    }

    // ---- LDS-DMA pieces of this wave (1 KB each): W1 slab = 40 pieces (K-tile kt, 8 rows), W2 slab = 20 pieces (16 rows)
    // (per-lane source offsets are rebuilt from two lane constants at every issue -- a handful of VALU ops -- instead of living in
    //  eight address registers across the slab loop: the kernel sits at the 256-VGPR limit)
    const int n2 = w < 4 ? 3 : 2, first2 = w < 4 ? 3 * w : 12 + 2 * (w - 4);
    const unsigned l1row = lane >> 3, l1kc = lane & 7, l2row = lane >> 2, l2kc = lane & 3;
    auto dma_w1 = [&](int slab, int buf) {
        const char* src = (const char*)(p.W1 + (size_t)slab * (2 * FF_HS * FF_C));   // wave-uniform
        unsigned lr = l1row, lk = l1kc;
        asm volatile("" : "+v"(lr), "+v"(lk));   // (not hoisted out of the slab loop)
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int piece = 5 * w + q;
            const unsigned row = (piece & 7) * 8 + lr;
            const unsigned off = (row * FF_C + (piece >> 3) * 64 + ((lk ^ (row & 7)) << 3)) * 2;
            glds16((const half_t*)(src + off), smem + W1B + buf * FF_W1_BYTES + piece * 1024);
        }
    };
    auto dma_w2 = [&](int slab, int buf) {
        const char* src = (const char*)(p.W2 + (size_t)slab * (FF_C * FF_HS));
        unsigned lr = l2row, lk = l2kc;
        asm volatile("" : "+v"(lr), "+v"(lk));
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < n2) {
                const unsigned n = (first2 + q) * 16 + lr;
                const unsigned off = (n * FF_HS + ((lk ^ ((0u - (n >> 2)) & 3)) << 3)) * 2;
                glds16((const half_t*)(src + off), smem + W2B + buf * FF_W2_BYTES + (first2 + q) * 1024);
            }
    };

    // ---- fragment addresses (bytes from lds0); re-derived from a laundered lane id inside the slab loop (see `addr` there)
    // W1 slab: five K-tiles [64 rows][128 B], 16-B chunk ^ (row & 7); this wave multiplies rows 32 half_ + 16 c + l15, c = 0 (h), 1 (gate)
    // W2 slab: [320 rows][64 B], 16-B chunk lq ^ (-(n >> 2) & 3); this wave's columns n = 160 half_ + 16 nf + l15
    // (the swizzle term depends on (n >> 2) & 3 only, which 16-column steps do not change: fragment nf sits at + nf * 1024)
    struct Addr { unsigned w1c0, w1c1, w2off, xw_own, xw_par, b1off; };
    auto make_addr = [&](int ln) {
        const int l15_ = ln & 15, lq_ = ln >> 4;
        Addr a;
        const unsigned w1row = (32 * half_ + l15_) * 128;
        a.w1c0 = w1row + (((0 * 4 + lq_) ^ (l15_ & 7)) << 4);
        a.w1c1 = w1row + (((1 * 4 + lq_) ^ (l15_ & 7)) << 4);
        const int n_lane = 160 * half_ + l15_;
        a.w2off = n_lane * 64 + ((lq_ ^ ((-(n_lane >> 2)) & 3)) << 4);
        a.xw_own = XB + ((w * 2) * 64 + ln) * 8;
        a.xw_par = XB + (((w ^ 1) * 2) * 64 + ln) * 8;
        a.b1off = B1B + (32 * half_ + 4 * lq_) * 2;
        return a;
    };

    dma_w1(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();

    const int rounds = (p.nstrips - (int)blockIdx.x * 4 + G * 4 - 1) / (G * 4);   // of this block (pair 0 has the most)
    for (int rd = 0; rd < rounds; ++rd) {
        const int strip = (rd * G + blockIdx.x) * 4 + pr;
        const bool live = strip < p.nstrips;
        const int row0 = (live ? strip : p.nstrips - 1) * 32;
        // the strip's activations: MFMA b-fragments [row fragment][k-step] (lane: token l15, k chunk lq), rows past M clamped
        h8 xf[2][10];
#pragma unroll
        for (int rf = 0; rf < 2; ++rf) {
            int m = row0 + rf * 16 + l15;
            m = m < p.M ? m : p.M - 1;
            const half_t* xp = p.X + (size_t)m * p.ldx + lq * 8;
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) xf[rf][ks] = *(const h8*)(xp + ks * 32);
        }
        f4 yacc[2][10];
#pragma unroll
        for (int rf = 0; rf < 2; ++rf)
#pragma unroll
            for (int nf = 0; nf < 10; ++nf) yacc[rf][nf] = (f4){0.f, 0.f, 0.f, 0.f};
        h4 hprev[2] = {(h4){0, 0, 0, 0}, (h4){0, 0, 0, 0}};   // own halves of the previous slab (phase B runs one step behind)
#pragma unroll
        for (int rf = 0; rf < 2; ++rf)
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) asm volatile("" : "+v"(xf[rf][ks]));   // complete BEFORE the loop (no vmcnt(0) inside it)

        for (int j = 0; j <= FF_NSLAB; ++j) {
#ifdef ANYV2V_EXPERIMENTS
            auto stamp = [&](int k) {
                if constexpr (KO == 7) {
                    if (blockIdx.x == 0 && rd == 1 && j >= 4 && j < 12 && (w & 3) == 0 && lane == 0)
                        g_ff_trace[((w >> 2) * 8 + (j - 4)) * 16 + k] = (long long)__builtin_amdgcn_s_memtime();
                }
            };
#else
            auto stamp = [&](int) {};
#endif
            stamp(0);
            // ---- the next slabs: W1(j + 1) (wrapping to the next round's slab 0) and W2(j), one step ahead of their use
            if (j < FF_NSLAB && KO != 1) {
                if (j + 1 < FF_NSLAB || rd + 1 < rounds) dma_w1(j + 1 < FF_NSLAB ? j + 1 : 0, (j + 1) & 1);
                dma_w2(j, j & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            stamp(1);
            int lane_j = lane;
            asm volatile("" : "+v"(lane_j));   // (address constants are NOT carried across the loop: the kernel sits at 256 VGPRs)
            const Addr ad = make_addr(lane_j);
            h4 hown[2];
            auto phase_a = [&]() {
                const unsigned wb = lds0 + W1B + (j & 1) * FF_W1_BYTES;
                const h4 bh = ffr64(lds0 + ad.b1off + j * 128), bg = ffr64(lds0 + ad.b1off + j * 128 + 32);
                f4 acc[2][2];
#pragma unroll
                for (int rf = 0; rf < 2; ++rf)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[rf][c] = (f4){0.f, 0.f, 0.f, 0.f};
                constexpr int D1 = (VAR & 1) ? 2 : 1;   // k-steps of read-ahead
                h8 wf[D1 + 1][2];   // ring over k-steps: the fragments of step ks + D1 are requested before step ks multiplies
                const unsigned wb0 = wb + ad.w1c0, wb1 = wb + ad.w1c1;
#define FF_RD_KS(KS)                                                                     \
    do {                                                                                 \
        wf[(KS) % (D1 + 1)][0] = ffr128(((KS) & 1) ? wb1 : wb0, ((KS) >> 1) * (64 * 128));          \
        wf[(KS) % (D1 + 1)][1] = ffr128(((KS) & 1) ? wb1 : wb0, ((KS) >> 1) * (64 * 128) + 2048);   \
    } while (0)
                FF_RD_KS(0);
                if constexpr (D1 == 2) FF_RD_KS(1);
                if constexpr ((VAR & 2) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 10; ++ks) {
                    if (ks + D1 < 10) FF_RD_KS(ks + D1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 2 < 10 && D1 == 2)
                        FF_LGKM(4);
                    else if (ks + 1 < 10)
                        FF_LGKM(2);
                    else
                        FF_LGKM(0);
#pragma unroll
                    for (int rf = 0; rf < 2; ++rf)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            if constexpr (KO == 4)
                                asm volatile("" ::"v"(wf[ks % (D1 + 1)][c]), "v"(xf[rf][ks]));
                            else
                                acc[rf][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks % (D1 + 1)][c], xf[rf][ks], acc[rf][c], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef FF_RD_KS
                if constexpr ((VAR & 2) != 0) __builtin_amdgcn_s_setprio(0);
                stamp(2);
#pragma unroll
                for (int rf = 0; rf < 2; ++rf)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        hown[rf][r] = KO == 2 ? (half_t)((acc[rf][0][r] + (float)bh[r]) * (acc[rf][1][r] + (float)bg[r]))
                                              : (half_t)((acc[rf][0][r] + (float)bh[r]) * av_gelu(acc[rf][1][r] + (float)bg[r]));
                ffw64(lds0 + ad.xw_own + (j & 1) * FF_X_BYTES, hown[0]);
                ffw64(lds0 + ad.xw_own + (j & 1) * FF_X_BYTES + 512, hown[1]);
                stamp(3);
            };
            auto phase_b = [&](int jb) {   // slab jb: own halves from registers (hprev), the partner's from LDS
                stamp(4);
                const unsigned wb = lds0 + W2B + (jb & 1) * FF_W2_BYTES;
                const unsigned xa = lds0 + ad.xw_par + (jb & 1) * FF_X_BYTES;
                h4 hp[2];
                hp[0] = KO == 6 ? hprev[0] : ffr64(xa);
                hp[1] = KO == 6 ? hprev[1] : ffr64(xa + 512);
                constexpr int D2 = (VAR & 1) ? 3 : 2;   // fragments of read-ahead
                h8 wf[D2 + 1];
                const unsigned wbl = wb + ad.w2off;
                wf[0] = ffr128(wbl, 0);
                wf[1] = ffr128(wbl, 1024);
                if constexpr (D2 == 3) wf[2] = ffr128(wbl, 2048);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (D2 == 3)
                    FF_LGKM(3);
                else
                    FF_LGKM(2);   // the partner's halves (older than the weight fragments)
                h8 hb[2];
#pragma unroll
                for (int rf = 0; rf < 2; ++rf)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hb[rf][e] = half_ ? hp[rf][e] : hprev[rf][e];        // slots 0..3: hidden 4 lq + e   (wave a's half)
                        hb[rf][4 + e] = half_ ? hprev[rf][e] : hp[rf][e];    // slots 4..7: hidden 16 + 4 lq + e (wave b's half)
                    }
                if constexpr ((VAR & 2) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int nf = 0; nf < 10; ++nf) {
                    if (nf + D2 < 10) wf[(nf + D2) % (D2 + 1)] = ffr128(wbl, (nf + D2) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                    if (nf + 3 < 10 && D2 == 3)
                        FF_LGKM(3);
                    else if (nf + 2 < 10)
                        FF_LGKM(2);
                    else if (nf + 1 < 10)
                        FF_LGKM(1);
                    else
                        FF_LGKM(0);
#pragma unroll
                    for (int rf = 0; rf < 2; ++rf) {
                        if constexpr (KO == 3)
                            asm volatile("" ::"v"(wf[nf % (D2 + 1)]), "v"(hb[rf]));
                        else
                            yacc[rf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nf % (D2 + 1)], hb[rf], yacc[rf][nf], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr ((VAR & 2) != 0) __builtin_amdgcn_s_setprio(0);
                stamp(5);
            };
            // ONE code sequence for both halves of the block; what differs is where the step's barrier sits.  Waves 0-3: A(j), B(j - 1),
            // barrier.  Waves 4-7 (their SIMD partners): A(j), barrier, B(j).  Between two barriers a SIMD therefore holds one wave in
            // [A(k), B(k - 1)] and one in [B(k - 1), A(k)]: the same work in opposite order (the pair partners w, w ^ 1 are in the same
            // half, so the exchange stays consistent).  (Two inlined copies of the phases in opposite order cost 40 spilled VGPRs.)
            auto step_sync = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                stamp(6);
                __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): this step's DMA pieces and exchange writes are complete
                stamp(7);
                if constexpr (KO != 5) __builtin_amdgcn_s_barrier();
                stamp(8);
                __builtin_amdgcn_sched_barrier(0);
            };
            if (j < FF_NSLAB) phase_a();
            if (late) {
                step_sync();
                hprev[0] = hown[0];
                hprev[1] = hown[1];
            }
            const int jb = j - 1 + late;
            if (jb >= 0 && jb < FF_NSLAB) phase_b(jb);
            if (!late) {
                hprev[0] = hown[0];
                hprev[1] = hown[1];
                step_sync();
            }
        }

        // ---- epilogue of the strip: + b2 -> fp16 -> (+ residual, fp16 add) -> 8-byte stores (a lane owns 4 consecutive channels)
        if (live) {
#pragma unroll
            for (int rf = 0; rf < 2; ++rf) {
                const int m = row0 + rf * 16 + l15;
                if (m < p.M) {
#pragma unroll
                    for (int nf = 0; nf < 10; ++nf) {
                        const int n = 160 * half_ + 16 * nf + 4 * lq;
                        const h4 b = *(const h4*)(smem + B2B + n * 2);
                        h4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (half_t)(yacc[rf][nf][r] + (float)b[r]);
                        if constexpr (RES) o = o + *(const h4*)(p.R + (size_t)m * p.ldr + n);
                        *(h4*)(p.Y + (size_t)m * p.ldy + n) = o;
                    }
                }
            }
        }
    }
}

extern "C" int anyv2v_ff_geglu_f16(const AnyV2VFFDesc* d, void* stream) {
    AV_CHECK(d != nullptr, "ff_geglu: null descriptor");
    AV_CHECK(d->X && d->W1 && d->b1 && d->W2 && d->b2 && d->Y, "ff_geglu: null X / W1 / b1 / W2 / b2 / Y");
    AV_CHECK(d->M > 0, "ff_geglu: bad M %d", d->M);
    if (d->C != FF_C || d->H != FF_H) {
        anyv2v_set_error("ff_geglu: only C = 320 with hidden 1280 is implemented (got C %d, hidden %d)", d->C, d->H);
        return ANYV2V_EUNSUPPORTED;
    }
    AV_CHECK(d->ldx % 8 == 0 && d->ldy % 4 == 0 && (d->R == nullptr || d->ldr % 4 == 0), "ff_geglu: ldx %% 8, ldy %% 4, ldr %% 4 must be 0");
    AV_CHECK(av_aligned16(d->X) && av_aligned16(d->W1) && av_aligned16(d->W2) && av_aligned16(d->b1) && av_aligned16(d->b2) &&
                 (((uintptr_t)d->Y) & 7) == 0 && (((uintptr_t)d->R) & 7) == 0,
             "ff_geglu: X / W1 / W2 / b1 / b2 must be 16-byte aligned, Y / R 8-byte aligned");
    FFK k;
    k.X = (const half_t*)d->X; k.W1 = (const half_t*)d->W1; k.b1 = (const half_t*)d->b1; k.W2 = (const half_t*)d->W2;
    k.b2 = (const half_t*)d->b2; k.R = (const half_t*)d->R; k.Y = (half_t*)d->Y;
    k.M = d->M; k.ldx = d->ldx; k.ldr = d->ldr; k.ldy = d->ldy;
    k.nstrips = (d->M + 31) / 32;
    const int blocks = (k.nstrips + 3) / 4;
    const dim3 grid(blocks < 256 ? blocks : 256);
#ifdef ANYV2V_EXPERIMENTS   // probe build only: knock-outs selected by flags bits 0-2 (results are wrong by construction)
    switch (d->flags & 7) {
#define AV_FF_KO(n) case n: hipLaunchKernelGGL((ff_fused_c320_kernel<true, n>), grid, dim3(512), 0, (hipStream_t)stream, k); return av_launch_status("ff_fused<KO>");
        AV_FF_KO(1) AV_FF_KO(2) AV_FF_KO(3) AV_FF_KO(4) AV_FF_KO(5) AV_FF_KO(6) AV_FF_KO(7)
#undef AV_FF_KO
        default: break;
    }
    switch ((d->flags >> 3) & 3) {   // variants (correct results): bit3 deeper fragment rings, bit4 s_setprio 1 around the MFMA loops
#define AV_FF_VAR(n) case (n == 0 ? 2 : n): hipLaunchKernelGGL((ff_fused_c320_kernel<true, 0, n>), grid, dim3(512), 0, (hipStream_t)stream, k); return av_launch_status("ff_fused<VAR>");
        AV_FF_VAR(1) AV_FF_VAR(0) AV_FF_VAR(3)
#undef AV_FF_VAR
        default: break;
    }
#endif
    if (d->R != nullptr)
        hipLaunchKernelGGL(ff_fused_c320_kernel<true>, grid, dim3(512), 0, (hipStream_t)stream, k);
    else
        hipLaunchKernelGGL(ff_fused_c320_kernel<false>, grid, dim3(512), 0, (hipStream_t)stream, k);
    return av_launch_status("ff_fused_c320");
}

#ifdef ANYV2V_EXPERIMENTS
extern "C" int anyv2v_ff_trace_read(void* host_dst) {   // probe build only (flags & 7 == 7 fills it)
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_ff_trace), sizeof(long long) * 2 * 8 * 16);
}
#endif
