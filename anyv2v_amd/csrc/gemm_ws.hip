// Weight-stationary streaming GEMM for the short-K Linear layers of the 64x64 level (K = 320; transformer_in: K = 512) (gfx950).
//
// Replaces (reference = TIGER-AI-Lab/AnyV2V): attn.to_q / to_k / to_v / to_out[0] at the 320-channel level
// (i2vgen-xl/pnp_utils.py:175,182-183,216), and the diffusers-0.26.3 Transformer2DModel / TransformerTemporalModel
// proj_in / proj_out and FeedForward GEGLU up-projection behind pipeline_i2vgen_xl.py:1146 -- the layers where a row of
// the token matrix is 640 bytes and the tile kernels of gemm.hip spend more time switching tiles than multiplying
// (profiles/r02_shape_report_B3.txt: 447-642 TF/s, HISTORY.md section 9).
//
// Structure:
//   * a block owns ONE 160-column slab of W for its whole life: W[160][320] = 100 KB sits in LDS (five [160][64]
//     K-tiles, 16-byte chunks XOR-swizzled by row & 7, filled once by LDS-DMA);
//   * every wave is autonomous: it walks 32-row strips of its block's row range and keeps the next five K-steps of
//     activations in REGISTERS (a ring of 10 x global_load_dwordx4 = 40 VGPRs, re-requested as soon as a piece has been
//     turned into MFMA fragments): 80 KB of activations in flight per CU, across strip boundaries, with no tile switch to
//     wait for;
//   * no barrier and no LDS write in the steady state (the activations never touch LDS), so the eight waves drift out of
//     phase and one wave's epilogue (convert / erf-GELU / LDS turn / residual / stores) runs under its SIMD partner's
//     MFMAs -- what the lock-step tile kernels cannot do;
//   * plain loads and stores only inside the loop: hipcc counts vmcnt / lgkmcnt itself (no LDS-DMA is in flight there).
// Slabs of one row range run on the same XCD at the same time (block -> (xcd, slab, range) below), so the activations
// come from HBM once and from that XCD's L2 for the other slabs.
#include "gemm_common.h"

namespace {
constexpr int WS_RW = 32;            // rows per wave strip
#ifndef WS_PRIO
#define WS_PRIO 1
#endif
}  // namespace

// TRACE (probe build only, tools/gemm_ws_trace.py): s_memtime stamps of the third strip of waves 0 and 4 of every block
// WS_K: reduction length (320: the 64x64 level's transformers; 512: transformer_in), WS_NS: W slab columns per block (W slab =
// WS_NS x WS_K x 2 bytes of LDS), WS_R: K-steps of register look-ahead (even: line halves requested in pairs), WQ: weight-fragment ring
template <int WS_K, int WS_NS, bool GEGLU, bool RES, int WS_R, int WQ, bool RR_EARLY, bool TRACE = false, bool LN = false>
__global__ __launch_bounds__(512) void gemm_ws_kernel(const GemmK p, const WsPlan plan) {
    constexpr int WS_KS = WS_K / 32;     // MFMA K-steps per strip
    constexpr int WS_NF = WS_NS / 16;    // 16-column MFMA fragments per slab
    constexpr int WS_W_BYTES = (WS_K / 64) * WS_NS * 128;
    constexpr int WS_BIAS_BYTES = WS_NS * 2 + (LN ? WS_NS * 4 : 0);   // slab bias (fp16) [+ the LayerNorm fold's column sums c1, fp32]
    static_assert(!(LN && RES), "the LayerNorm-folded form has no residual epilogue (QKV / to_q / GEGLU take none)");
    constexpr int OUT_W = GEGLU ? WS_NS / 2 : WS_NS;   // output columns of a slab
    constexpr int SLAB_LD = OUT_W + 8;                 // halves; keeps rows 16-byte aligned, breaks the power-of-2 stride
    constexpr int SLAB_BYTES = 16 * SLAB_LD * 2;       // per wave: one 16-row half strip
    constexpr int CPRW = OUT_W / 8;                    // 16-byte chunks per output row
    constexpr int NIT = (16 * CPRW + 63) / 64;         // store iterations per 16-row half strip
    __shared__ __attribute__((aligned(16))) char smem[WS_W_BYTES + WS_BIAS_BYTES + 8 * SLAB_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    // block -> (xcd, slab, row range): the S slab blocks of a row range share an XCD (blockIdx % 8, observed placement --
    // speed only) and start together, so A is fetched from HBM once per range and from L2 by the other slabs
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (idx >= plan.px * plan.S) return;
    const int slab = idx % plan.S;
    const int range = xcd * plan.px + idx / plan.S;
    const int s_begin = range * plan.spr;
    const int s_end = s_begin + plan.spr < plan.nstrips ? s_begin + plan.spr : plan.nstrips;
    const int n_wave = slab * WS_NS;

    // Activations are loaded as [16 rows][32 k] pieces, lane = (row l >> 2, 16-byte chunk l & 3): a quad of lanes reads 64
    // contiguous bytes (the fragment-shaped form -- lane = (row l & 15, chunk l >> 4), 16 different cache lines per quad pass
    // -- was bound by the texture addresser: ~130 cycles per load, profiles/r03_gemm_ws_ab_v1_fragment_loads.txt) and turned
    // into the MFMA B-operand layout (token l & 15, k-chunk l >> 4) by four ds_bpermute_b32 per fragment, one K-step ahead.
    auto a_row = [&](int strip, int mf) {
        int row = strip * WS_RW + mf * 16 + (lane >> 2);
        row = row < p.M ? row : p.M - 1;
        return p.A0 + (size_t)row * p.lda0 + (lane & 3) * 8;
    };
    const int bperm_addr = ((lane & 15) * 4 + (lane >> 4)) * 4;   // source lane (row l & 15, chunk l >> 4), in bytes
    auto to_frag = [&](const h8& raw) {
        typedef int i4 __attribute__((ext_vector_type(4)));
        const i4 r = __builtin_bit_cast(i4, raw);
        i4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_ds_bpermute(bperm_addr, r[i]);
        return __builtin_bit_cast(h8, o);
    };
    // Register ring: WS_R K-steps of raw activation pieces; the slot of step t is re-requested for step t + WS_R (the same
    // strip's, or the wave's next strip's) as soon as its ds_bpermute reads have been issued.
    h8 a[WS_R][2];
    int strip = s_begin + w;
    const half_t *pc0, *pc1;   // this strip's rows (mf = 0 / 1), then the next strip's
    {   // first strip: requested before the W slab, lands while the DMA runs
        const int s0 = strip < s_end ? strip : (s_begin < plan.nstrips ? s_begin : 0);
        pc0 = a_row(s0, 0);
        pc1 = a_row(s0, 1);
#pragma unroll
        for (int s = 0; s < WS_R; ++s) {
            a[s][0] = *(const h8*)(pc0 + s * 32);
            a[s][1] = *(const h8*)(pc1 + s * 32);
        }
    }
    {   // W slab -> LDS: 100 pieces of 1 KB (8 rows x 128 B of one K-tile), source-side swizzle, destination lane-linear
        const half_t* wslab = p.W + (size_t)n_wave * WS_K;
        for (int j = w; j < (WS_K / 64) * (WS_NS / 8); j += 8) {
            const int kt = j / (WS_NS / 8), rg = j - kt * (WS_NS / 8);
            const int row = rg * 8 + (lane >> 3), kc = (lane & 7) ^ (row & 7);
            glds16(wslab + (size_t)row * WS_K + kt * 64 + kc * 8, smem + kt * (WS_NS * 128) + rg * 1024);
        }
        if (tid < WS_NS / 4) {
            const h4 b = *(const h4*)(p.bias != nullptr ? p.bias + n_wave + tid * 4 : p.zeros);
            *(h4*)(smem + WS_W_BYTES + tid * 8) = b;
            if constexpr (LN) *(f4*)(smem + WS_W_BYTES + WS_NS * 2 + tid * 16) = *(const f4*)(p.ln_c1 + n_wave + tid * 4);
        }
    }
    __syncthreads();  // (hipcc drains the LDS-DMA with vmcnt(0) here; nothing LDS-bound is in flight afterwards)

    const char* const wl = smem + l15 * 128;
    const int wsw[2] = {((0 * 4 + lq) ^ (l15 & 7)) * 16, ((1 * 4 + lq) ^ (l15 & 7)) * 16};
    half_t* const slabp = (half_t*)(smem + WS_W_BYTES + WS_BIAS_BYTES + w * SLAB_BYTES);
    const half_t* const bias_l = (const half_t*)(smem + WS_W_BYTES);
    const int n_out_wave = GEGLU ? n_wave / 2 : n_wave;
    static_assert(WS_KS % WS_R == 0 && WS_K % 64 == 0 && WS_NS % 16 == 0 && (!GEGLU || WS_NS % 32 == 0), "static ring indices need WS_R | WS_KS");
    static_assert(WS_W_BYTES + WS_BIAS_BYTES + 8 * SLAB_BYTES <= 160 * 1024, "W slab + epilogue slabs must fit in LDS");

    int nth = 0;
    if constexpr (TRACE) if (w >= plan.trace_waves) return;   // probe: one wave per SIMD (4) or per pair of SIMDs (2) only
    while (strip < s_end) {
        long long* tr = nullptr;
        if constexpr (TRACE) {
            if (nth == 2 && (w == 0 || w == (plan.trace_waves > 4 ? 4 : 1)) && lane == 0) tr = p.trace + ((size_t)blockIdx.x * 2 + (w != 0)) * 16;
            ++nth;
            if (tr) tr[0] = (long long)__builtin_amdgcn_s_memtime();
        }
        const int next = strip + 8;
        const int pre = next < s_end ? next : strip;   // last strip of the wave: harmless re-read of its own rows
        const half_t* const pn0 = a_row(pre, 0);
        const half_t* const pn1 = a_row(pre, 1);
        const int m_wave = strip * WS_RW;
        // Residual rows of both halves: requested either first (RR_EARLY: the whole K loop to arrive, 40 registers held
        // through it) or after the K loop.  vmcnt retires in order, so waiting for them also waits for every older load --
        // in the late form that is the whole look-ahead issued before them.  Rows past M are clamped, never stored.
        h8 rr[2][RES ? NIT : 1];
        auto res_request = [&]() {
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int it = 0; it < (RES ? NIT : 0); ++it) {
                    const int c = it * 64 + lane;
                    const int row = c / CPRW, cc = c - row * CPRW;
                    int m = m_wave + mf * 16 + row;
                    m = m < p.M ? m : p.M - 1;
                    rr[mf][it] = *(const h8*)(p.R + (size_t)m * p.ldr + n_out_wave + cc * 8);
                }
        };
        if constexpr (RES && RR_EARLY) res_request();
        f4 acc[2][WS_NF];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WS_NF; ++j) {   // the accumulators start at the bias (lane: channels 16 j + 4 lq .. + 3)
                const h4 b = *(const h4*)(bias_l + j * 16 + 4 * lq);
                acc[i][j] = LN ? (f4){0.f, 0.f, 0.f, 0.f} : (f4){(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
            }
        // LayerNorm fold (LN): the rows arrive UN-normalised; with W' = W diag(gamma), c1[n] = sum_k W'[n][k], b' = b + W beta:
        //   LN(x) W^T + b = rstd (x W'^T - mean c1) + b'.   The row statistics come out of the matrix pipe as well: one MFMA of the
        // activation fragment with itself per K-step (Gram matrix: its diagonal is sum x^2) and one with a fragment of ones (sum x).
        f4 accq[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}}, accs[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
        h8 ones;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (half_t)1.0f;

        // K loop, written in issue order and pinned (hipcc otherwise sinks the look-ahead loads behind the last MFMA and
        // reads each weight fragment right in front of its consumers): weight fragments roll WQ - 1 MFMA pairs ahead;
        // during K-step s the raw pieces of step s + 1 are turned into fragments (ds_bpermute) and their registers
        // re-requested for step s + 1 + WS_R.
        constexpr int NFR = WS_KS * WS_NF;  // weight fragment reads per strip
        auto wfrag = [&](int idx) {
            const int s = idx / WS_NF, nf = idx - s * WS_NF;
            return *(const h8*)(wl + (s >> 1) * (WS_NS * 128) + wsw[s & 1] + nf * 2048);
        };
        // K-step t of this strip (t < WS_KS) or t - WS_KS of the next one, into its ring slot.  WS_R = 10 requests steps in
        // (even, odd) pairs: the two 64-byte halves of every 128-byte line back to back, so the second one finds the line
        // in L1 / in flight (one step apart it has been evicted again: 2 x the L2 requests, QKV at 196608 rows 160 -> 204 us)
        auto request = [&](int t, int n) {
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int i = 0; i < n; ++i) {
                    const int u = t + i;
                    const half_t* q = u < WS_KS ? (mf ? pc1 : pc0) : (mf ? pn1 : pn0);
                    a[u % WS_R][mf] = *(const h8*)(q + (u < WS_KS ? u : u - WS_KS) * 32);
                }
        };
        constexpr bool PAIRS = WS_R % 2 == 0;
        h8 wq[WQ], fr[2][2];
#pragma unroll
        for (int i = 0; i < WQ - 1; ++i) wq[i] = wfrag(i);
        fr[0][0] = to_frag(a[0][0]);
        fr[0][1] = to_frag(a[0][1]);
        if (!PAIRS) request(WS_R, 1);
        // the SIMD's other wave is usually in its epilogue (a dense VALU stream) while this one multiplies: MFMAs first
        if (WS_PRIO) __builtin_amdgcn_s_setprio(WS_PRIO);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < WS_KS; ++s)
#pragma unroll
        for (int nf = 0; nf < WS_NF; ++nf) {   // (two nested fully unrolled loops: one flat loop of 128 iterations is only partly unrolled)
            const int idx = s * WS_NF + nf;
            if (idx + WQ - 1 < NFR) wq[(idx + WQ - 1) % WQ] = wfrag(idx + WQ - 1);
            acc[0][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[idx % WQ], fr[s & 1][0], acc[0][nf], 0, 0, 0);
            acc[1][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[idx % WQ], fr[s & 1][1], acc[1][nf], 0, 0, 0);
            if constexpr (LN) {
                if (nf == 4 || nf == 5) {   // row statistics of this K-step, row block nf - 4
                    accq[nf - 4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[s & 1][nf - 4], fr[s & 1][nf - 4], accq[nf - 4], 0, 0, 0);
                    accs[nf - 4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, fr[s & 1][nf - 4], accs[nf - 4], 0, 0, 0);
                }
            }
            if (nf == 2 && s + 1 < WS_KS) {   // next K-step's fragments, seven MFMA pairs ahead of their first use
                fr[(s + 1) & 1][0] = to_frag(a[(s + 1) % WS_R][0]);
                fr[(s + 1) & 1][1] = to_frag(a[(s + 1) % WS_R][1]);
                if (!PAIRS) request(s + 1 + WS_R, 1);
                if (PAIRS && ((s + 1) & 1)) request(s + WS_R, 2);
            }
            if constexpr (TRACE) if (tr && nf == WS_NF - 1) tr[1 + s] = (long long)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (WS_PRIO) __builtin_amdgcn_s_setprio(0);
        pc0 = pn0;
        pc1 = pn1;
        if constexpr (RES && !RR_EARLY) res_request();

        // ---- wave-private epilogue: two 16-row halves, (GEGLU) -> fp16 -> LDS turn -> (+residual) -> 16-byte stores
        // the epilogue's lane-derived offsets must not be hoisted out of the strip loop (they would live across the K loop
        // next to ~200 accumulator / fragment registers and spill): launder the lane id once per strip
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int l15_e = lane_e & 15, lq_e = lane_e >> 4;
        float ln_mean[2] = {0.f, 0.f}, ln_rstd[2] = {1.f, 1.f};
        if constexpr (LN) {
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                // D[i][j] of the Gram MFMA sits in lane (j = l15, lq = i >> 2), register i & 3: token j's own sum x^2 is register
                // l15 & 3 of the lane with lq == l15 >> 2 -- select, then fetch it from that lane; sum x is in every register
                const int r3 = l15_e & 3;
                float d = r3 == 0 ? accq[mf][0] : (r3 == 1 ? accq[mf][1] : (r3 == 2 ? accq[mf][2] : accq[mf][3]));
                d = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((l15_e >> 2) * 16 + l15_e) * 4, __builtin_bit_cast(int, d)));
                const float mean = accs[mf][0] * (1.0f / WS_K);
                const float var = fmaxf(d * (1.0f / WS_K) - mean * mean, 0.0f);
                ln_mean[mf] = mean;
                ln_rstd[mf] = __builtin_amdgcn_rsqf(var + p.ln_eps);
            }
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int nf = 0; nf < WS_NF; ++nf) {
                    const f4 c1 = *(const f4*)((const char*)bias_l + WS_NS * 2 + (nf * 16 + 4 * lq_e) * 4);
                    const h4 b = *(const h4*)(bias_l + nf * 16 + 4 * lq_e);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[mf][nf][r] = fmaf(fmaf(-ln_mean[mf], c1[r], acc[mf][nf][r]), ln_rstd[mf], (float)b[r]);
                }
        }
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            if constexpr (GEGLU) {
#pragma unroll
                for (int np = 0; np < WS_NF / 2; ++np) {
                    h4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o[r] = (half_t)(acc[mf][2 * np][r] * av_gelu(acc[mf][2 * np + 1][r]));  // fp32 throughout, one rounding
                    *(h4*)(slabp + l15_e * SLAB_LD + np * 16 + 4 * lq_e) = o;
                }
            } else {
#pragma unroll
                for (int nf = 0; nf < WS_NF; ++nf) {
                    h4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)acc[mf][nf][r];
                    *(h4*)(slabp + l15_e * SLAB_LD + nf * 16 + 4 * lq_e) = o;
                }
            }
            if (RES && mf == 0) {   // ONE wait for all residual rows, before the first store: a later wait for a load would
                                    // make hipcc drain the stores issued in between (loads and stores share vmcnt)
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(rr[g][RES ? it : 0]));
            }
            // same-wave LDS traffic is ordered; the compiler inserts the lgkmcnt wait for the read-back
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = it * 64 + lane_e;
                const int row = c / CPRW, cc = c - row * CPRW;
                const bool ok = (16 * CPRW % 64 == 0 || c < 16 * CPRW) && m_wave + mf * 16 + row < p.M;
                h8 v = *(const h8*)(slabp + (ok ? row * SLAB_LD + cc * 8 : 0));
                if constexpr (RES) v = v + rr[mf][it];  // fp16 add: correctly rounded, == the fp32 add + rounding of two fp16 values
                if (ok) *(h8*)(p.C + (size_t)(m_wave + mf * 16 + row) * p.ldc + n_out_wave + cc * 8) = v;
            }
        }
        if constexpr (TRACE) if (tr) tr[11] = (long long)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        strip = next;
    }
}

// host side: can this launch run on the weight-stationary kernel, and how are (slab, row range) dealt to the 256 blocks
static int ws_slab_cols(const AnyV2VGemmDesc* d) {   // 0 = shape not covered
    if (d->C0 == 320) return 160;
    // K = 512 (transformer_in): GEGLU with 128-column slabs (1291 -> 834 us at 196608 rows); plain launches would need 64-column
    // slabs (128 + their epilogue slabs exceed LDS), which only pays for N = 512 (x 1.07-1.16; QKV N = 1536: x 0.87-0.96, left to
    // the tile kernels) -- profiles/r03_gemm_ws_ab.txt
    if (d->C0 == 512) return d->act == ACT_GEGLU ? 128 : (d->N == 512 ? 64 : 0);
    return 0;
}

bool av_gemm_ws_eligible(const AnyV2VGemmDesc* d) {
    const int ns = ws_slab_cols(d);
    if (d->ln_c1 != nullptr && (d->R != nullptr || (d->C0 == 512 && d->act != ACT_GEGLU))) return false;
    return d->mode == MODE_LINEAR && ns > 0 && d->C1 == 0 && d->N % ns == 0 && d->N / ns <= 32 &&
           (d->act == ACT_NONE || (d->act == ACT_GEGLU && d->R == nullptr)) && d->rowvec == nullptr && d->M > 0;
}

int av_gemm_ws_launch(const GemmK& k_in, const AnyV2VGemmDesc* d, hipStream_t s) {
    GemmK k = k_in;
    WsPlan plan;
    const int ns = ws_slab_cols(d);
    plan.S = d->N / ns;
    plan.px = 32 / plan.S;
    plan.nstrips = (d->M + WS_RW - 1) / WS_RW;
    const int nranges = 8 * plan.px;
    plan.spr = (plan.nstrips + nranges - 1) / nranges;
    plan.trace_waves = (d->flags & 8192) ? 4 : ((d->flags & 16384) ? 1 : 8);
    const bool geglu = d->act == ACT_GEGLU, res = d->R != nullptr;
#define WS_GO(K, NS, G, RS, R, Q, E, T) hipLaunchKernelGGL((gemm_ws_kernel<K, NS, G, RS, R, Q, E, T>), dim3(256), dim3(512), 0, s, k, plan)
#ifdef ANYV2V_EXPERIMENTS  // probe build only: per-strip phase timestamps (flags bit5), tools/gemm_ws_trace.py
    if ((d->flags & 32) && d->C0 == 320 && d->workspace != nullptr && (size_t)256 * 2 * 16 * sizeof(long long) <= (size_t)d->workspace_bytes) {
        k.trace = (long long*)d->workspace;
        if (geglu) WS_GO(320, 160, true, false, 5, 8, false, true);
        else if (res) WS_GO(320, 160, false, true, 10, 4, false, true);
        else WS_GO(320, 160, false, false, 10, 4, false, true);
        return av_launch_status("gemm_ws<trace>");
    }
#endif
    // (A/B in profiles/r03_gemm_ws_ab.txt: ring depth 5 vs 10, weight look-ahead 3 vs 7, residual requested first vs after the
    //  K loop -- all within 3 % except: pairs matter for QKV at 196608 rows, the late residual request for the +residual launches)
    if (d->ln_c1 != nullptr) {   // LayerNorm folded in (QKV / to_q / GEGLU-up of the transformer blocks)
        k.ln_c1 = d->ln_c1;
        k.ln_eps = d->ln_eps;
#define WS_GO_LN(K, NS, G, R, Q) hipLaunchKernelGGL((gemm_ws_kernel<K, NS, G, false, R, Q, false, false, true>), dim3(256), dim3(512), 0, s, k, plan)
        if (d->C0 == 320) {
            if (geglu) WS_GO_LN(320, 160, true, 5, 4);
            else WS_GO_LN(320, 160, false, 10, 4);
        } else {
            WS_GO_LN(512, 128, true, 8, 4);
        }
#undef WS_GO_LN
        return av_launch_status("gemm_ws<LN>");
    }
    if (d->C0 == 320) {
        if (geglu) WS_GO(320, 160, true, false, 5, 8, false, false);
        else if (res) WS_GO(320, 160, false, true, 10, 4, false, false);
        else WS_GO(320, 160, false, false, 10, 4, false, false);
    } else {
        if (geglu) WS_GO(512, 128, true, false, 8, 8, false, false);
        else if (res) WS_GO(512, 64, false, true, 8, 4, false, false);
        else WS_GO(512, 64, false, false, 8, 8, false, false);
    }
#undef WS_GO
    return av_launch_status("gemm_ws");
}
