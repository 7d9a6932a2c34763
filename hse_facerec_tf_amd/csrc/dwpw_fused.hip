// Fused depthwise 3x3 (+scale+shift+ReLU6) -> pointwise 1x1 (+shift+ReLU6), NHWC fp32, gfx950.
//
// Replaces one whole MobileNet block of the frozen graph (e.g. nodes #35-#49: DepthwiseConv2dNative, Mul, Add,
// Relu, Minimum, Maximum, Conv2D 1x1, Add, Relu, Minimum, Maximum) for the EARLY blocks (C = 32 / 64), which are
// HBM-bound in both halves: unfused, the depthwise output is written to HBM and read straight back by the
// pointwise GEMM (2 x 1.18 MB / 2 x 0.59 MB per face at 192x192).  Here it never leaves the CU:
//
//   phase 1  the workgroup's 256 threads compute the depthwise result of an 8 x 16 pixel patch (128 GEMM rows) with
//            the sliding-window scheme of dwconv.hip (coalesced float4 loads from clamped addresses, padding folded
//            into weights / row factors, rows requested two iterations ahead) and write it as the A operand into LDS
//            in the swizzled 128-B-row K-tile layout of pwconv_f32.hip;
//   phase 2  the 4 waves (2x2) run the fp32 MFMA GEMM of that tile against the pointwise weights, which stay resident
//            in LDS for the life of the (persistent) workgroup, add the shift, apply ReLU6 and store full 128-B rows.
//
// All output channels sit in one tile (BN = Cout), so the depthwise work is done exactly once per pixel.  Phases of
// the 2-4 co-resident workgroups of a CU interleave, which is what overlaps the streaming with the MFMAs.
// HBM traffic per patch: (8s+2) x (16s+2) x C in (halo re-reads hit L2) + 128 x Cout out.
#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct DwPwParams {
    const float4* x;       // [N,H,W,C]
    const float4* wd;      // depthwise [9][C/4]
    const float4* dscale;  // [C/4]
    const float4* dshift;  // [C/4]
    const float* wp;       // pointwise, transposed [Cout][C]
    const float* pshift;   // [Cout]
    float* y;              // [N,OH,OW,Cout]
    int H, W, OH, OW, pad_t, pad_l, tiles_w, tiles_h;
    unsigned total;        // N * tiles_h * tiles_w patches
    int reverse;           // sweep direction (common.h)
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * 32 + 4 * (chunk ^ ((row >> 1) & 7)); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

// ---- phase 1: depthwise result of patch (n, oh0.., ow0..) -> A operand tile in LDS; `ptid` in [0,256) ----------------
template <int STRIDE, int C>
__device__ __forceinline__ void dw_phase(const DwPwParams& p, float (*As)[128 * 32], int ptid, int n, int oh0, int ow0) {
    constexpr int C4 = C / 4, TW = 16, TH = 8;
    constexpr int U = TW * C4;                 // (column, channel-quad) work items per patch row: 128 or 256
    constexpr int ROWS_PER_THREAD = U >= 256 ? TH : TH / 2;
    static_assert(U == 128 || U == 256, "C must be 32 or 64");
    const int u = ptid % U;
    const int tw = u / C4, c4 = u % C4;
    const int row0 = (U >= 256) ? 0 : (ptid / U) * ROWS_PER_THREAD;
    const float4 dsc = p.dscale[c4], dsh = p.dshift[c4];
    const int ow = min(ow0 + tw, p.OW - 1);            // clamped: out-of-range columns are computed, never stored
    const int iw0 = ow * STRIDE - p.pad_l;
    const float ml = iw0 >= 0 ? 1.f : 0.f, mm = (iw0 + 1 >= 0 && iw0 + 1 < p.W) ? 1.f : 0.f, mr = iw0 + 2 < p.W ? 1.f : 0.f;
    float4 wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float m = (i % 3 == 0) ? ml : (i % 3 == 1 ? mm : mr);
        const float4 wr = p.wd[i * C4 + c4];   // 9 L1-resident loads per patch; keeps 36 VGPRs free
        wk[i] = make_float4(wr.x * m, wr.y * m, wr.z * m, wr.w * m);
    }
    const int cl = max(iw0, 0) * C4, cm = min(max(iw0 + 1, 0), p.W - 1) * C4, cr = min(iw0 + 2, p.W - 1) * C4;
    const float4* xin = p.x + (size_t)n * p.H * p.W * C4 + c4;
    struct Row { float4 l, m, r; float k; };
    auto load_row = [&](int ih) {
        Row q;
        const int ihc = min(max(ih, 0), p.H - 1);
        const float4* row = xin + (size_t)ihc * p.W * C4;
        q.l = row[cl]; q.m = row[cm]; q.r = row[cr];
        q.k = (ih >= 0 && ih < p.H) ? 1.f : 0.f;
        return q;
    };
    auto row_sum = [&](const Row& q, int b) {
        float4 a = make_float4(q.l.x * wk[b].x, q.l.y * wk[b].y, q.l.z * wk[b].z, q.l.w * wk[b].w);
        a = fma4(q.m, wk[b + 1], a);
        return fma4(q.r, wk[b + 2], a);
    };
    auto emit = [&](int th, const Row& a, const Row& b, const Row& c) {
        const float4 sa = row_sum(a, 0), sb = row_sum(b, 3), sc = row_sum(c, 6);
        float4 acc = make_float4(sa.x * a.k, sa.y * a.k, sa.z * a.k, sa.w * a.k);
        acc = make_float4(fmaf(sb.x, b.k, acc.x), fmaf(sb.y, b.k, acc.y), fmaf(sb.z, b.k, acc.z), fmaf(sb.w, b.k, acc.w));
        acc = make_float4(fmaf(sc.x, c.k, acc.x), fmaf(sc.y, c.k, acc.y), fmaf(sc.z, c.k, acc.z), fmaf(sc.w, c.k, acc.w));
        const float4 o = fma4(acc, dsc, dsh);
        f32x4 v;
        v[0] = relu6(o.x); v[1] = relu6(o.y); v[2] = relu6(o.z); v[3] = relu6(o.w);
        *(f32x4*)(&As[c4 >> 3][swz(th * TW + tw, c4 & 7)]) = v;
    };
    const int ohb = oh0 + row0;
    if (STRIDE == 1) {
        const int ih = ohb - p.pad_t;
        Row r0 = load_row(ih), r1 = load_row(ih + 1), r2 = load_row(ih + 2), r3 = load_row(ih + 3);
#pragma unroll
        for (int j = 0; j < ROWS_PER_THREAD; ++j) {
            const Row r4 = load_row(ih + j + 4);
            emit(row0 + j, r0, r1, r2);
            r0 = r1; r1 = r2; r2 = r3; r3 = r4;
        }
    } else {
        const int ih = ohb * 2 - p.pad_t;
        Row r0 = load_row(ih), r1 = load_row(ih + 1), r2 = load_row(ih + 2);
#pragma unroll
        for (int j = 0; j < ROWS_PER_THREAD; ++j) {
            const Row n1 = load_row(ih + 2 * j + 3), n2 = load_row(ih + 2 * j + 4);
            emit(row0 + j, r0, r1, r2);
            r0 = r2; r1 = n1; r2 = n2;
        }
    }
}

// ---- phase 2: [128 x C] . [C x BN] on the fp32 MFMA + shift + ReLU6 + store; `wave4` in [0,4), lane in [0,64) ---------
template <int C, int BN>
__device__ __forceinline__ void mfma_phase(const DwPwParams& p, const float (*As)[128 * 32], const float (*Bs)[BN * 32], int wave4,
                                           int lane, int n, int oh0, int ow0) {
    constexpr int KT = C / 32, TW = 16, TH = 8, WN = BN / 2, NI = WN / 32;
    const int wm = wave4 >> 1, wn = wave4 & 1;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 a[2], b[NI];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = *(const f32x4*)(&As[kt][swz(wm * 64 + mi * 32 + li, 2 * s + lh)]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = *(const f32x4*)(&Bs[kt][swz(wn * WN + ni * 32 + li, 2 * s + lh)]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
        }
    float psh[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) psh[ni] = p.pshift[wn * WN + ni * 32 + li];
    // epilogue: tile row R = th*16 + tw; accumulator register r of lane-half lh -> R = base + (r&3) + 8*(r>>2) + 4*lh
    // (full patches store unconditionally: a per-store bounds branch costs an s_waitcnt vmcnt(0) per store)
    const bool full = oh0 + TH <= p.OH && ow0 + TW <= p.OW;
    float* ybase = p.y + ((size_t)n * p.OH * p.OW) * BN;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = wn * WN + ni * 32 + li;
            const int Rb = wm * 64 + mi * 32 + 4 * lh;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int R = Rb + (r & 3) + 8 * (r >> 2);
                    ybase[((size_t)(oh0 + (R >> 4)) * p.OW + ow0 + (R & 15)) * BN + col] = relu6(acc[mi][ni][r] + psh[ni]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int R = Rb + (r & 3) + 8 * (r >> 2);
                    const int oh = oh0 + (R >> 4), ow = ow0 + (R & 15);
                    if (oh < p.OH && ow < p.OW) ybase[((size_t)oh * p.OW + ow) * BN + col] = relu6(acc[mi][ni][r] + psh[ni]);
                }
            }
        }
}

__device__ __forceinline__ void decode_patch(const DwPwParams& p, unsigned t, int& n, int& oh0, int& ow0) {
    const unsigned lt = xcd_remap_dir(t, p.total, p.reverse);
    ow0 = (lt % p.tiles_w) * 16;
    oh0 = ((lt / p.tiles_w) % p.tiles_h) * 8;
    n = lt / (p.tiles_w * p.tiles_h);
}

template <int C, int BN>
__device__ __forceinline__ void load_pw_weights(const DwPwParams& p, float (*Bs)[BN * 32], int tid, int nthreads) {
    constexpr int KT = C / 32;
    for (int i = tid; i < KT * BN * 8; i += nthreads) {
        const int ch = i & 7, r = (i >> 3) % BN, kt = i / (BN * 8);
        *(f32x4*)(&Bs[kt][swz(r, ch)]) = *(const f32x4*)(p.wp + (size_t)r * C + kt * 32 + ch * 4);
    }
}

// Variant A: 256 threads, the two phases alternate inside each workgroup; co-resident workgroups overlap them.
template <int STRIDE, int C, int BN, int OCC>
__global__ __launch_bounds__(256, OCC) void dwpw_fused_kernel(DwPwParams p) {
    constexpr int KT = C / 32;
    __shared__ __attribute__((aligned(16))) float As[KT][128 * 32];
    __shared__ __attribute__((aligned(16))) float Bs[KT][BN * 32];
    const int tid = threadIdx.x;
    load_pw_weights<C, BN>(p, Bs, tid, 256);
    for (unsigned t = blockIdx.x; t < p.total; t += gridDim.x) {
        int n, oh0, ow0;
        decode_patch(p, t, n, oh0, ow0);
        __syncthreads();   // previous tile's MFMA reads of As are done (first pass: Bs is written)
        dw_phase<STRIDE, C>(p, As, tid, n, oh0, ow0);
        __syncthreads();
        mfma_phase<C, BN>(p, As, Bs, tid >> 6, tid & 63, n, oh0, ow0);
    }
}

template <int STRIDE, int C, int BN, int OCC>
int launch_t(const DwPwParams& p, hipStream_t s) {
    // (a producer/consumer-wave form of this kernel -- 512 threads, double-buffered A tile -- was no faster than the
    // alternating phases at 4 WG/CU and was retired in round 2; git history has it)
    const unsigned cap = 256u * OCC;
    const unsigned g = p.total < cap ? p.total : cap;
    HSEFR_LAUNCH((dwpw_fused_kernel<STRIDE, C, BN, OCC>), dim3(g), dim3(256), 0, s, p);
    return launch_status("dwpw_fused");
}

}  // namespace


bool dwpw_fused_supported(int c, int cout, int stride, int act_dw, int act_pw) {
    return (c == 32 || c == 64) && (cout == 64 || cout == 128) && (stride == 1 || stride == 2) &&
           act_dw == HSEFR_ACT_RELU6 && act_pw == HSEFR_ACT_RELU6;
}

int launch_dwpw_fused(const float* x, const float* wd, const float* dscale, const float* dshift, const float* wp_t,
                      const float* pshift, float* y, int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh,
                      int ow, int cout, int act_dw, int act_pw, hipStream_t s) {
    HSEFR_REQUIRE(dwpw_fused_supported(c, cout, stride, act_dw, act_pw), HSEFR_ERR_UNSUPPORTED,
                  "dwpw_fused: c=%d cout=%d stride=%d acts=%d/%d not covered (c in {32,64}, cout in {64,128}, ReLU6)", c, cout,
                  stride, act_dw, act_pw);
    HSEFR_REQUIRE(n >= 0 && h > 0 && w > 0 && oh > 0 && ow > 0, HSEFR_ERR_INVALID, "dwpw_fused: bad shape");
    if (n == 0) return HSEFR_OK;
    DwPwParams p;
    p.x = (const float4*)x; p.wd = (const float4*)wd; p.dscale = (const float4*)dscale; p.dshift = (const float4*)dshift;
    p.wp = wp_t; p.pshift = pshift; p.y = y;
    p.H = h; p.W = w; p.OH = oh; p.OW = ow; p.pad_t = pad_t; p.pad_l = pad_l;
    p.tiles_w = (ow + 15) / 16; p.tiles_h = (oh + 7) / 8;
    const long long total = (long long)n * p.tiles_w * p.tiles_h;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "dwpw_fused: grid too large");
    p.total = (unsigned)total;
    p.reverse = sweep_reverse();
#define HSEFR_DWPW(S, CC, NN, O) return launch_t<S, CC, NN, O>(p, s)
    if (stride == 1) {
        if (c == 32 && cout == 64) HSEFR_DWPW(1, 32, 64, 4);
        if (c == 32 && cout == 128) HSEFR_DWPW(1, 32, 128, 3);
        if (c == 64 && cout == 64) HSEFR_DWPW(1, 64, 64, 3);
        HSEFR_DWPW(1, 64, 128, 2);
    } else {
        if (c == 32 && cout == 64) HSEFR_DWPW(2, 32, 64, 4);
        if (c == 32 && cout == 128) HSEFR_DWPW(2, 32, 128, 3);
        if (c == 64 && cout == 64) HSEFR_DWPW(2, 64, 64, 3);
        HSEFR_DWPW(2, 64, 128, 2);
    }
#undef HSEFR_DWPW
}

}  // namespace hsefr
