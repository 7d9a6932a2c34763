// First convolution of the trunk: dense 3x3 conv over the 3-channel image + shift + ReLU6,
// NHWC fp32, gfx950.
//
// Replaces graph nodes #30-34 (Conv2D k=[3,3,3,32] strides [1,2,2,1] SAME with the BN scale
// pre-folded into the kernel -> Add shift -> Relu -> Minimum 6 -> Maximum 0), run by
// tf_sess.run at facerec_test.py:120 / facial_analysis.py:109.
//
// K = 27 is too shallow for MFMA and the layer is ~HBM-bound (AI 9.8 flop/B: it writes 2.7x
// what it reads), so this is a VALU kernel shaped like the depthwise one:
//   * a thread owns 4 output channels of ONE output column and walks down a strip of rows
//     with its 27 float4 weights resident in VGPRs (108 FMAs per output, no LDS);
//   * lanes = (column, channel-quad) -> the float4 store of a wave is 1 KiB contiguous in
//     NHWC; the 8 lanes of a column read the same 9 input floats per row (one broadcast
//     request), and with stride 2 one of the three input rows is carried in registers.
// Algorithmic bytes per image: 4*(3*H*W + Cout*OH*OW) + 4*28*Cout.
#include "common.h"

namespace hsefr {

namespace {

struct C3Params {
    const float* x;
    const float4* w;      // [9*3][Q] float4: TF HWIO [3,3,3,Cout] viewed as float4 over Cout
    const float4* shift;  // [Q]
    float4* y;
    int H, W, Q, OH, OW, pad_t, pad_l, TH, tiles_h, tiles_x;
    unsigned nwg;
};

struct Px3 { float c[3]; };

template <int STRIDE, int ACT>
__global__ __launch_bounds__(256) void conv3x3_c3_kernel(C3Params p) {
    const unsigned bid = xcd_remap(blockIdx.x, p.nwg);
    const int tx = bid % p.tiles_x;
    const int th = (bid / p.tiles_x) % p.tiles_h;
    const int n = bid / (p.tiles_x * p.tiles_h);
    const int t = tx * 256 + threadIdx.x;
    if (t >= p.OW * p.Q) return;
    const int ow = t / p.Q;
    const int q = t - ow * p.Q;

    float4 wk[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) wk[i] = p.w[i * p.Q + q];
    const float4 sh = p.shift[q];

    const int iw0 = ow * STRIDE - p.pad_l;
    const bool ok0 = iw0 >= 0, ok1 = (iw0 + 1 >= 0) && (iw0 + 1 < p.W), ok2 = iw0 + 2 < p.W;
    const float* xin = p.x + (size_t)n * p.H * p.W * 3;

    auto load_row = [&](int ih, Px3* r) {
        const bool okh = ih >= 0 && ih < p.H;
        const float* row = xin + ((long long)(okh ? ih : 0) * p.W + iw0) * 3;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            r[0].c[ci] = (okh && ok0) ? row[ci] : 0.f;
            r[1].c[ci] = (okh && ok1) ? row[3 + ci] : 0.f;
            r[2].c[ci] = (okh && ok2) ? row[6 + ci] : 0.f;
        }
    };
    auto accum_row = [&](const Px3* r, int dy, float4& acc) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = r[dx].c[ci];
                const float4 wv = wk[(dy * 3 + dx) * 3 + ci];
                acc.x = fmaf(v, wv.x, acc.x);
                acc.y = fmaf(v, wv.y, acc.y);
                acc.z = fmaf(v, wv.z, acc.z);
                acc.w = fmaf(v, wv.w, acc.w);
            }
    };

    const int oh0 = th * p.TH;
    const int oh1 = min(oh0 + p.TH, p.OH);
    float4* yout = p.y + ((size_t)n * p.OH * p.OW + ow) * p.Q + q;

    Px3 r0[3], r1[3], r2[3];
    if (STRIDE == 1) {
        load_row(oh0 - p.pad_t, r0);
        load_row(oh0 - p.pad_t + 1, r1);
    } else {
        load_row(oh0 * 2 - p.pad_t, r0);
    }
    for (int oh = oh0; oh < oh1; ++oh) {
        if (STRIDE == 1) {
            load_row(oh - p.pad_t + 2, r2);
        } else {
            load_row(oh * 2 - p.pad_t + 1, r1);
            load_row(oh * 2 - p.pad_t + 2, r2);
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        accum_row(r0, 0, acc);
        accum_row(r1, 1, acc);
        accum_row(r2, 2, acc);
        float4 o;
        o.x = apply_act<ACT>(acc.x + sh.x);
        o.y = apply_act<ACT>(acc.y + sh.y);
        o.z = apply_act<ACT>(acc.z + sh.z);
        o.w = apply_act<ACT>(acc.w + sh.w);
        yout[(size_t)oh * p.OW * p.Q] = o;
        if (STRIDE == 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { r0[i] = r1[i]; r1[i] = r2[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) r0[i] = r2[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// im2col + fp32 MFMA variant (Cout = 32 or 64).  The VALU kernel above issues 18 dword loads per
// 108 FMAs and is bound by address processing (measured 2.0 TB/s of algorithmic traffic); here a
// thread gathers its output pixel's 3 x 9 contiguous input floats with 9 dwordx3 loads, writes one
// 128-B LDS row [k = dy*9 + dx*3 + ci, zero-padded to 32], and the 256 x 32 x Cout product runs on
// v_mfma_f32_32x32x2_f32 (exact fp32).  Persistent workgroups; the gather of the next 256-pixel
// tile is in flight during the MFMAs and the stores of the current one.  Same LDS row swizzle
// and K-permutation as the pointwise GEMM (pwconv_f32.hip).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };

__device__ __forceinline__ int swz32(int row, int chunk) { return row * 32 + 4 * (chunk ^ ((row >> 1) & 7)); }

struct C3MParams {
    const float* x;
    const float* w;      // TF HWIO [3,3,3,Cout]
    const float* shift;  // [Cout]
    float* y;
    int H, W, OH, OW, pad_t, pad_l, cout;
    long long P;         // n * OH * OW output pixels
    unsigned tiles;
};

template <int STRIDE, int NI, int ACT>
__global__ __launch_bounds__(256, 4) void conv3x3_c3_mfma_kernel(C3MParams p) {
    constexpr int BM = 256;
    __shared__ __attribute__((aligned(16))) float As[BM * 32];
    __shared__ __attribute__((aligned(16))) float Bs[NI * 32 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // weights -> LDS rows [n][k], k = (dy*3+dx)*3+ci, zero for k >= 27
    for (int i = tid; i < NI * 32 * 32; i += 256) {
        const int n = i >> 5, k = i & 31;
        Bs[swz32(n, k >> 2) + (k & 3)] = k < 27 ? p.w[k * p.cout + n] : 0.f;
    }

    // Branch-free gather: clamped (always valid) addresses, padding applied as 0/1 factors.
    F3 g[9];
    float mk[9];
    auto gather = [&](unsigned tile) {
        unsigned pix = tile * BM + tid;
        if (pix > (unsigned)(p.P - 1)) pix = (unsigned)(p.P - 1);
        const unsigned ow = pix % (unsigned)p.OW;
        const unsigned t2 = pix / (unsigned)p.OW;
        const unsigned oh = t2 % (unsigned)p.OH;
        const unsigned n = t2 / (unsigned)p.OH;
        const int iw0 = (int)ow * STRIDE - p.pad_l, ih0 = (int)oh * STRIDE - p.pad_t;
        const float* img = p.x + (size_t)n * p.H * p.W * 3;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ih = ih0 + dy, ihc = min(max(ih, 0), p.H - 1);
            const float my = (ih >= 0 && ih < p.H) ? 1.f : 0.f;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int iw = iw0 + dx, iwc = min(max(iw, 0), p.W - 1);
                g[dy * 3 + dx] = *(const F3*)(img + ((size_t)ihc * p.W + iwc) * 3);
                mk[dy * 3 + dx] = (iw >= 0 && iw < p.W) ? my : 0.f;
            }
        }
    };
    auto scatter = [&]() {
        float v[32];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            v[3 * q] = g[q].a * mk[q];
            v[3 * q + 1] = g[q].b * mk[q];
            v[3 * q + 2] = g[q].c * mk[q];
        }
#pragma unroll
        for (int q = 27; q < 32; ++q) v[q] = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            f32x4 o;
            o[0] = v[4 * c]; o[1] = v[4 * c + 1]; o[2] = v[4 * c + 2]; o[3] = v[4 * c + 3];
            *(f32x4*)(&As[swz32(tid, c)]) = o;
        }
    };

    unsigned t = blockIdx.x;
    if (t >= p.tiles) return;
    gather(t);
    scatter();
    __syncthreads();
    float sh[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) sh[ni] = p.shift[ni * 32 + li];

    while (true) {
        const unsigned tn = t + gridDim.x;
        const bool more = tn < p.tiles;
        if (more) gather(tn);
        f32x16 acc[2][NI];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 a[2], b[NI];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = *(const f32x4*)(&As[swz32(wave * 64 + mi * 32 + li, 2 * s + lh)]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = *(const f32x4*)(&Bs[swz32(ni * 32 + li, 2 * s + lh)]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();                    // every wave is done reading As
        if (more) scatter();
        const unsigned pbase = t * BM + wave * 64 + 4 * lh;
        const bool full_tile = (unsigned long long)t * BM + BM <= (unsigned long long)p.P;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                float* yp = p.y + (size_t)(pbase + mi * 32) * p.cout + ni * 32 + li;
                if (full_tile) {   // unconditional stores: no per-store branch, no per-store vmcnt(0)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        yp[(size_t)((r & 3) + 8 * (r >> 2)) * p.cout] = apply_act<ACT>(acc[mi][ni][r] + sh[ni]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned dr = (r & 3) + 8 * (r >> 2);
                        if (pbase + mi * 32 + dr < (unsigned)p.P) yp[(size_t)dr * p.cout] = apply_act<ACT>(acc[mi][ni][r] + sh[ni]);
                    }
                }
            }
        if (!more) break;
        __syncthreads();                    // next tile's rows are in As
        t = tn;
    }
}

template <int STRIDE, int NI>
int launch_mfma(const C3MParams& p, int act, hipStream_t s) {
    const unsigned g = p.tiles < 1024u ? p.tiles : 1024u;
    dim3 grid(g), block(256);
#define HSEFR_C3M(A) HSEFR_LAUNCH((conv3x3_c3_mfma_kernel<STRIDE, NI, A>), grid, block, 0, s, p)
    if (act == HSEFR_ACT_RELU6) HSEFR_C3M(HSEFR_ACT_RELU6);
    else if (act == HSEFR_ACT_RELU) HSEFR_C3M(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_NONE) HSEFR_C3M(HSEFR_ACT_NONE);
    else { set_error("conv_c3: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_C3M
    return launch_status("conv_c3_mfma");
}

HSEFR_KNOB(g_c3_impl, 0);  // dev builds: 0 = auto, 1 = VALU kernel, 2 = MFMA kernel

}  // namespace

#ifdef HSEFR_DEV
void set_c3_impl(int v) { g_c3_impl = v; }
#endif

int launch_conv_c3(const float* x, const float* wgt, const float* shift, float* y, int n, int h, int w,
                   int kh, int kw, int stride, int pad_t, int pad_l, int oh, int ow, int cout, int act,
                   hipStream_t s) {
    HSEFR_REQUIRE(kh == 3 && kw == 3, HSEFR_ERR_UNSUPPORTED, "conv_c3: only 3x3 kernels (got %dx%d)", kh, kw);
    HSEFR_REQUIRE(cout % 4 == 0 && cout > 0, HSEFR_ERR_UNSUPPORTED, "conv_c3: cout=%d must be a multiple of 4", cout);
    HSEFR_REQUIRE(stride == 1 || stride == 2, HSEFR_ERR_UNSUPPORTED, "conv_c3: stride %d", stride);
    HSEFR_REQUIRE(n >= 0 && h > 0 && w > 0 && oh > 0 && ow > 0, HSEFR_ERR_INVALID, "conv_c3: bad shape");
    if (n == 0) return HSEFR_OK;
    const bool mfma_ok = (cout == 32 || cout == 64) && h >= 3 && w >= 3;
    if (mfma_ok && g_c3_impl != 1) {
        C3MParams q;
        q.x = x; q.w = wgt; q.shift = shift; q.y = y;
        q.H = h; q.W = w; q.OH = oh; q.OW = ow; q.pad_t = pad_t; q.pad_l = pad_l; q.cout = cout;
        q.P = (long long)n * oh * ow;
        const long long tiles = (q.P + 255) / 256;
        HSEFR_REQUIRE(q.P < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv_c3: too many output pixels");
        q.tiles = (unsigned)tiles;
        if (stride == 2) return cout == 32 ? launch_mfma<2, 1>(q, act, s) : launch_mfma<2, 2>(q, act, s);
        return cout == 32 ? launch_mfma<1, 1>(q, act, s) : launch_mfma<1, 2>(q, act, s);
    }
    C3Params p;
    p.x = x; p.w = (const float4*)wgt; p.shift = (const float4*)shift; p.y = (float4*)y;
    p.H = h; p.W = w; p.Q = cout / 4; p.OH = oh; p.OW = ow; p.pad_t = pad_t; p.pad_l = pad_l;
    p.tiles_x = (ow * p.Q + 255) / 256;
    int th = oh;
    while (th > 4 && (long long)n * p.tiles_x * ((oh + th - 1) / th) < 1024) th = (th + 1) / 2;
    p.TH = th;
    p.tiles_h = (oh + th - 1) / th;
    const long long nwg = (long long)n * p.tiles_x * p.tiles_h;
    HSEFR_REQUIRE(nwg < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv_c3: grid too large");
    p.nwg = (unsigned)nwg;
    dim3 grid((unsigned)nwg), block(256);
#define HSEFR_C3_LAUNCH(S, A) HSEFR_LAUNCH((conv3x3_c3_kernel<S, A>), grid, block, 0, s, p)
    if (stride == 1) {
        if (act == HSEFR_ACT_RELU6) HSEFR_C3_LAUNCH(1, HSEFR_ACT_RELU6);
        else if (act == HSEFR_ACT_RELU) HSEFR_C3_LAUNCH(1, HSEFR_ACT_RELU);
        else if (act == HSEFR_ACT_NONE) HSEFR_C3_LAUNCH(1, HSEFR_ACT_NONE);
        else { set_error("conv_c3: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
    } else {
        if (act == HSEFR_ACT_RELU6) HSEFR_C3_LAUNCH(2, HSEFR_ACT_RELU6);
        else if (act == HSEFR_ACT_RELU) HSEFR_C3_LAUNCH(2, HSEFR_ACT_RELU);
        else if (act == HSEFR_ACT_NONE) HSEFR_C3_LAUNCH(2, HSEFR_ACT_NONE);
        else { set_error("conv_c3: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
    }
#undef HSEFR_C3_LAUNCH
    return launch_status("conv_c3");
}

}  // namespace hsefr
