// Definitions shared by the GEMM translation units (gemm.hip, gemm_ws.hip): kernel argument block, gather addressing.
#pragma once
#include "common.h"

enum { MODE_LINEAR = 0, MODE_CONV2D = 1, MODE_TEMPORAL = 2 };
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2, ACT_GEGLU = 3, ACT_F32OUT = 4 };


struct GemmK {
    const half_t* A0;
    const half_t* A1;
    const half_t* W;
    half_t* C;
    const half_t* bias;
    const half_t* rowvec;
    const half_t* R;
    const half_t* zeros;
    int M, N, C0, C1, lda0, lda1, ldc, ldr, ldrv, rowvec_div;
    int mode, Hi, Wi, Ho, Wo, stride, up, F, HW, act;
    int pad_lo;  // conv2d: zero rows / columns before the first pixel (1 = "same" 3x3; 0 = pad only right / bottom)
    int taps, Ktot, nt0, nt1, tilesN;
    int vec_epi;  // bias / rowvec may be read as 8-byte vectors
    int splits;   // split-K factor (128-row kernel only): each split writes an fp32 partial tile, reduced afterwards
    float* partial;  // [splits][M][N] fp32 workspace
    long long* trace;  // debug (flags bit5): 32 timestamps per block, see tools/gemm_trace.py
    const float* ln_c1;  // LayerNorm fold (gemm_ws.hip): column sums of the gamma-scaled weights, or nullptr
    float ln_eps;
    // persistent kernel, rastered tile order (0 = classic): an XCD round covers rast_gm x rast_gn output tiles; rast_sm x rast_sn
    // super-tiles, walked M-fastest (rast_nfast = 0) or N-fastest
    int rast_gm, rast_gn, rast_sm, rast_sn, rast_nfast;
};

struct RowInfo {
    int base;  // linear/temporal: row m (or -1); conv2d: img * Hi * Wi
    int y, x;  // conv2d: yo*stride-1, xo*stride-1 ; temporal: y = frame index
};

template <int MODE>
__device__ __forceinline__ RowInfo make_row(const GemmK& p, int m) {
    RowInfo r;
    const bool ok = m < p.M;
    if constexpr (MODE == MODE_CONV2D) {
        const int hw = p.Ho * p.Wo;
        const int img = m / hw, rem = m - img * hw;
        const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
        r.base = img * p.Hi * p.Wi;
        r.y = ok ? yo * p.stride - p.pad_lo : -(1 << 28);
        r.x = xo * p.stride - p.pad_lo;
    } else if constexpr (MODE == MODE_TEMPORAL) {
        r.base = m;
        r.y = ok ? (m / p.HW) % p.F : -(1 << 28);
        r.x = 0;
    } else {
        r.base = ok ? m : -1;
        r.y = r.x = 0;
    }
    return r;
}

// source row of output row `r` for filter tap `tap`, or -1 when the tap falls into the zero padding (branch-free)
template <int MODE>
__device__ __forceinline__ int src_row(const GemmK& p, const RowInfo& r, int tap) {
    if constexpr (MODE == MODE_CONV2D) {
        const int dy = tap / 3, dx = tap - dy * 3;
        int yi = r.y + dy, xi = r.x + dx;
        const int ly = p.Hi << p.up, lx = p.Wi << p.up;
        const bool ok = (yi >= 0) & (yi < ly) & (xi >= 0) & (xi < lx);
        yi >>= p.up;
        xi >>= p.up;
        return ok ? r.base + yi * p.Wi + xi : -1;
    } else if constexpr (MODE == MODE_TEMPORAL) {
        const int f = r.y + tap - 1;
        const bool ok = (f >= 0) & (f < p.F);
        return ok ? r.base + (tap - 1) * p.HW : -1;
    } else {
        return r.base;
    }
}

// per-K-tile A source: wave-uniform (base pointer, leading dim, column offset) + per-row select against the zero line
struct ASrc {
    const half_t* base;
    int ld;
};
__device__ __forceinline__ ASrc a_source(const GemmK& p, int kt_c, int kc) {
    ASrc s;
    const bool first = kt_c < p.nt0;
    s.base = (first ? p.A0 + kt_c * 64 : p.A1 + (kt_c - p.nt0) * 64) + kc * 8;
    s.ld = first ? p.lda0 : p.lda1;
    return s;
}
__device__ __forceinline__ const half_t* a_addr(const GemmK& p, const ASrc& s, int sr) {
    const half_t* g = s.base + (long long)sr * s.ld;
    return sr < 0 ? p.zeros : g;
}

__device__ __forceinline__ void glds16(const half_t* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}


// ---- weight-stationary kernel (gemm_ws.hip): launch plan + host entry points used by gemm.hip's dispatch ----
struct WsPlan {
    int S;        // 160-column W slabs (= blocks per row range)
    int px;       // row ranges per XCD (px * S <= 32 blocks of the 32 CUs of an XCD)
    int nstrips;  // 32-row strips in M
    int spr;      // strips per row range
    int trace_waves;  // probe build: waves of a block that work (8; 4 / 1 = one wave per SIMD / per CU), tools/gemm_ws_trace.py
};
bool av_gemm_ws_eligible(const AnyV2VGemmDesc* d);
int av_gemm_ws_launch(const GemmK& k, const AnyV2VGemmDesc* d, hipStream_t s);

// ---- one-wave-per-SIMD persistent kernel (gemm_sw.hip): 192 x 320 tiles, 4 waves, direct 16-byte stores ----
bool av_gemm_sw_eligible(const AnyV2VGemmDesc* d);
int av_gemm_sw_launch(GemmK& k, const AnyV2VGemmDesc* d, hipStream_t s);
// stream-K form (flags bit26 allows it, bit27 forces it): blocks to launch (0 = do not take it), its workspace need, the launch
int av_gemm_sw_sk_blocks(const AnyV2VGemmDesc* d, bool force);
size_t av_gemm_sw_sk_workspace(int blocks);
int av_gemm_sw_sk_launch(GemmK& k, const AnyV2VGemmDesc* d, int blocks, hipStream_t s);

// ---- 3x3 convolution with the A operand reused from LDS across the dx taps (gemm_swh.hip): flags bit28 takes it where eligible ----
bool av_gemm_swh_eligible(const AnyV2VGemmDesc* d);
int av_gemm_swh_launch(GemmK& k, const AnyV2VGemmDesc* d, hipStream_t s);
