// Weight-stationary streaming GEMM for the K = 320 Linear layers of the 64x64 level (gfx950).
//
// Replaces (reference = TIGER-AI-Lab/AnyV2V): attn.to_q / to_k / to_v / to_out[0] at the 320-channel level
// (i2vgen-xl/pnp_utils.py:175,182-183,216), and the diffusers-0.26.3 Transformer2DModel / TransformerTemporalModel
// proj_in / proj_out and FeedForward GEGLU up-projection behind pipeline_i2vgen_xl.py:1146 -- the layers where a row of
// the token matrix is 640 bytes and the tile kernels of gemm.hip spend more time switching tiles than multiplying
// (profiles/r02_shape_report_B3.txt: 447-642 TF/s).
//
// Structure ("weight-stationary kernel"):
//   * a block owns ONE 160-column slab of W for its whole life: W[160][320] = 100 KB sits in LDS (five [160][64]
//     K-tiles, 16-byte chunks XOR-swizzled by row & 7, filled once by LDS-DMA);
//   * every wave is autonomous: it walks 32-row strips of its block's row range, holds a strip's activations as MFMA
//     fragments in REGISTERS (20 x global_load_dwordx4 per strip = 80 VGPRs) and re-requests a fragment register for the
//     NEXT strip right after its last use, i.e. a whole strip (2.5-5 us of work) ahead: 160 KB of activations in flight
//     per CU, across strip boundaries, with no tile switch to wait for;
//   * no barrier and no LDS write in the steady state (the activations never touch LDS), so the eight waves drift out of
//     phase and one wave's epilogue (convert / erf-GELU / LDS turn / residual / stores) runs under its SIMD partner's
//     MFMAs -- what the lock-step tile kernels cannot do;
//   * plain loads and stores only inside the loop: hipcc counts vmcnt / lgkmcnt itself (no LDS-DMA is in flight there).
// Slabs of one row range run on the same XCD at the same time (block -> (xcd, slab, range) below), so the activations
// come from HBM once and from that XCD's L2 for the other slabs.
#include "gemm_common.h"

namespace {

constexpr int WS_K = 320;            // reduction length this kernel is built for
constexpr int WS_KS = WS_K / 32;     // MFMA K-steps per strip
constexpr int WS_NS = 160;           // W slab columns per block
constexpr int WS_NF = WS_NS / 16;    // 16-column MFMA fragments per slab
constexpr int WS_RW = 32;            // rows per wave strip
#ifndef WS_PRIO
#define WS_PRIO 1
#endif
constexpr int WS_EARLY = 8;          // K-steps whose fragment registers are re-requested inside the K loop (the rest: after the epilogue)
constexpr int WS_W_BYTES = (WS_K / 64) * WS_NS * 128;  // 102400
constexpr int WS_BIAS_BYTES = WS_NS * 2;               // slab bias, fp16
}  // namespace

// TRACE (probe build only, tools/gemm_ws_trace.py): s_memtime stamps of the third strip of waves 0 and 4 of every block
template <bool GEGLU, bool RES, bool TRACE = false>
__global__ __launch_bounds__(512) void gemm_ws_kernel(const GemmK p, const WsPlan plan) {
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
    h8 a[WS_KS][2];
    int strip = s_begin + w;
    {   // first strip: requested before the W slab, lands while the DMA runs
        const int s0 = strip < s_end ? strip : (s_begin < plan.nstrips ? s_begin : 0);
        const half_t* p0 = a_row(s0, 0);
        const half_t* p1 = a_row(s0, 1);
#pragma unroll
        for (int s = 0; s < WS_KS; ++s) {
            a[s][0] = *(const h8*)(p0 + s * 32);
            a[s][1] = *(const h8*)(p1 + s * 32);
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
        }
    }
    __syncthreads();  // (hipcc drains the LDS-DMA with vmcnt(0) here; nothing LDS-bound is in flight afterwards)

    const char* const wl = smem + l15 * 128;
    const int wsw[2] = {((0 * 4 + lq) ^ (l15 & 7)) * 16, ((1 * 4 + lq) ^ (l15 & 7)) * 16};
    half_t* const slabp = (half_t*)(smem + WS_W_BYTES + WS_BIAS_BYTES + w * SLAB_BYTES);
    const half_t* const bias_l = (const half_t*)(smem + WS_W_BYTES);
    const int n_out_wave = GEGLU ? n_wave / 2 : n_wave;

    int nth = 0;
    while (strip < s_end) {
        long long* tr = nullptr;
        if constexpr (TRACE) {
            if (nth == 2 && (w == 0 || w == 4) && lane == 0) tr = p.trace + ((size_t)blockIdx.x * 2 + (w >> 2)) * 16;
            ++nth;
            if (tr) tr[0] = (long long)__builtin_amdgcn_s_memtime();
        }
        const int next = strip + 8;
        const int pre = next < s_end ? next : strip;   // last strip of the wave: harmless re-read of its own rows
        const half_t* const pn0 = a_row(pre, 0);
        const half_t* const pn1 = a_row(pre, 1);
        f4 acc[2][WS_NF];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WS_NF; ++j) {   // the accumulators start at the bias (lane: channels 16 j + 4 lq .. + 3)
                const h4 b = *(const h4*)(bias_l + j * 16 + 4 * lq);
                acc[i][j] = (f4){(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
            }

        // K loop, written in issue order and pinned (hipcc otherwise sinks the next strip's loads behind the last MFMA and
        // reads each weight fragment right in front of its consumers): weight fragments roll three ahead of their MFMA
        // pair; the fragment registers of K-steps 0 .. WS_EARLY-1 are re-requested for the wave's next strip as soon as the
        // step is done (a whole strip ahead).  The last K-steps are re-requested AFTER the epilogue instead (still
        // WS_EARLY steps ahead of their use): their registers are free during the epilogue -- the residual rows are loaded
        // into them -- and, vmcnt retiring in order, the residual wait then only covers loads that are several steps old.
        constexpr int NFR = WS_KS * WS_NF;  // weight fragment reads per strip
        auto wfrag = [&](int idx) {
            const int s = idx / WS_NF, nf = idx - s * WS_NF;
            return *(const h8*)(wl + (s >> 1) * (WS_NS * 128) + wsw[s & 1] + nf * 2048);
        };
        h8 wq[4], fr[2][2];
        wq[0] = wfrag(0);
        wq[1] = wfrag(1);
        wq[2] = wfrag(2);
        fr[0][0] = to_frag(a[0][0]);
        fr[0][1] = to_frag(a[0][1]);
        // the SIMD's other wave is usually in its epilogue (a dense VALU stream) while this one multiplies: MFMAs first
        if (WS_PRIO) __builtin_amdgcn_s_setprio(WS_PRIO);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int idx = 0; idx < NFR; ++idx) {
            const int s = idx / WS_NF, nf = idx - s * WS_NF;
            if (idx + 3 < NFR) wq[(idx + 3) & 3] = wfrag(idx + 3);
            acc[0][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[idx & 3], fr[s & 1][0], acc[0][nf], 0, 0, 0);
            acc[1][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[idx & 3], fr[s & 1][1], acc[1][nf], 0, 0, 0);
            if (nf == 2 && s + 1 < WS_KS) {   // next K-step's fragments, seven MFMA pairs ahead of their first use
                fr[(s + 1) & 1][0] = to_frag(a[s + 1][0]);
                fr[(s + 1) & 1][1] = to_frag(a[s + 1][1]);
            }
            if (nf == WS_NF - 1 && (s & 1) && s < WS_EARLY) {   // both halves of a 128-byte line back to back
                a[s - 1][0] = *(const h8*)(pn0 + (s - 1) * 32);
                a[s][0] = *(const h8*)(pn0 + s * 32);
                a[s - 1][1] = *(const h8*)(pn1 + (s - 1) * 32);
                a[s][1] = *(const h8*)(pn1 + s * 32);
            }
            if constexpr (TRACE) if (tr && nf == WS_NF - 1) tr[1 + s] = (long long)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
        }

        if (WS_PRIO) __builtin_amdgcn_s_setprio(0);
        // ---- wave-private epilogue: two 16-row halves, (+bias | GEGLU) -> fp16 -> LDS turn -> (+residual) -> 16-byte stores
        // the epilogue's lane-derived offsets must not be hoisted out of the strip loop (they would live across the K loop
        // next to 176 accumulator / fragment registers and spill): launder the lane id once per strip
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int l15_e = lane_e & 15, lq_e = lane_e >> 4;
        const int m_wave = strip * WS_RW;
        // Residual rows of both halves are requested up front (their registers: the fragment registers of the last K-steps,
        // free until after the epilogue) and waited for ONCE, before the first store: any later wait for a load would make
        // hipcc drain the stores issued in between (loads and stores share vmcnt).  Rows past M are clamped, never stored.
        h8 rr[2][RES ? NIT : 1];
        if constexpr (RES) {
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int c = it * 64 + lane_e;
                    const int row = c / CPRW, cc = c - row * CPRW;
                    int m = m_wave + mf * 16 + row;
                    m = m < p.M ? m : p.M - 1;
                    rr[mf][it] = *(const h8*)(p.R + (size_t)m * p.ldr + n_out_wave + cc * 8);
                }
        }
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            if constexpr (GEGLU) {
#pragma unroll
                for (int np = 0; np < WS_NF / 2; ++np) {
                    h4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[r] = (half_t)(acc[mf][2 * np][r] * av_gelu(acc[mf][2 * np + 1][r]));  // fp32 throughout, one rounding
                    }
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
            if (RES && mf == 0) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(rr[g][it]));
            }
            // same-wave LDS traffic is ordered; the compiler inserts the lgkmcnt wait for the read-back
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = it * 64 + lane_e;
                const int row = c / CPRW, cc = c - row * CPRW;
                const bool ok = (16 * CPRW % 64 == 0 || c < 16 * CPRW) && m_wave + mf * 16 + row < p.M;
                h8 v = *(const h8*)(slabp + (ok ? row * SLAB_LD + cc * 8 : 0));
                if constexpr (RES) {
                    v = v + rr[mf][it];  // fp16 add: correctly rounded, == the fp32 add + rounding of two fp16 values
                }
                if (ok) *(h8*)(p.C + (size_t)(m_wave + mf * 16 + row) * p.ldc + n_out_wave + cc * 8) = v;
            }
        }
        if constexpr (TRACE) if (tr) tr[11] = (long long)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = WS_EARLY; s < WS_KS; ++s) a[s][0] = *(const h8*)(pn0 + s * 32);
#pragma unroll
        for (int s = WS_EARLY; s < WS_KS; ++s) a[s][1] = *(const h8*)(pn1 + s * 32);
        __builtin_amdgcn_sched_barrier(0);
        strip = next;
    }
}

// host side: can this launch run on the weight-stationary kernel, and how are (slab, row range) dealt to the 256 blocks
bool av_gemm_ws_eligible(const AnyV2VGemmDesc* d) {
    return d->mode == MODE_LINEAR && d->C0 == WS_K && d->C1 == 0 && d->N % WS_NS == 0 && d->N / WS_NS <= 32 &&
           (d->act == ACT_NONE || (d->act == ACT_GEGLU && d->R == nullptr)) && d->rowvec == nullptr && d->M > 0;
}

int av_gemm_ws_launch(const GemmK& k_in, const AnyV2VGemmDesc* d, hipStream_t s) {
    GemmK k = k_in;
    WsPlan plan;
    plan.S = d->N / WS_NS;
    plan.px = 32 / plan.S;
    plan.nstrips = (d->M + WS_RW - 1) / WS_RW;
    const int nranges = 8 * plan.px;
    plan.spr = (plan.nstrips + nranges - 1) / nranges;
#ifdef ANYV2V_EXPERIMENTS  // probe build only: per-strip phase timestamps (flags bit5), tools/gemm_ws_trace.py
    if ((d->flags & 32) && d->workspace != nullptr && (size_t)256 * 2 * 16 * sizeof(long long) <= (size_t)d->workspace_bytes) {
        k.trace = (long long*)d->workspace;
        if (d->act == ACT_GEGLU)
            hipLaunchKernelGGL((gemm_ws_kernel<true, false, true>), dim3(256), dim3(512), 0, s, k, plan);
        else if (d->R != nullptr)
            hipLaunchKernelGGL((gemm_ws_kernel<false, true, true>), dim3(256), dim3(512), 0, s, k, plan);
        else
            hipLaunchKernelGGL((gemm_ws_kernel<false, false, true>), dim3(256), dim3(512), 0, s, k, plan);
        return av_launch_status("gemm_ws<trace>");
    }
#endif
    if (d->act == ACT_GEGLU)
        hipLaunchKernelGGL((gemm_ws_kernel<true, false>), dim3(256), dim3(512), 0, s, k, plan);
    else if (d->R != nullptr)
        hipLaunchKernelGGL((gemm_ws_kernel<false, true>), dim3(256), dim3(512), 0, s, k, plan);
    else
        hipLaunchKernelGGL((gemm_ws_kernel<false, false>), dim3(256), dim3(512), 0, s, k, plan);
    return av_launch_status("gemm_ws");
}
