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

}  // namespace

int launch_conv_c3(const float* x, const float* wgt, const float* shift, float* y, int n, int h, int w,
                   int kh, int kw, int stride, int pad_t, int pad_l, int oh, int ow, int cout, int act,
                   hipStream_t s) {
    HSEFR_REQUIRE(kh == 3 && kw == 3, HSEFR_ERR_UNSUPPORTED, "conv_c3: only 3x3 kernels (got %dx%d)", kh, kw);
    HSEFR_REQUIRE(cout % 4 == 0 && cout > 0, HSEFR_ERR_UNSUPPORTED, "conv_c3: cout=%d must be a multiple of 4", cout);
    HSEFR_REQUIRE(stride == 1 || stride == 2, HSEFR_ERR_UNSUPPORTED, "conv_c3: stride %d", stride);
    HSEFR_REQUIRE(n >= 0 && h > 0 && w > 0 && oh > 0 && ow > 0, HSEFR_ERR_INVALID, "conv_c3: bad shape");
    if (n == 0) return HSEFR_OK;
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
#define HSEFR_C3_LAUNCH(S, A) hipLaunchKernelGGL((conv3x3_c3_kernel<S, A>), grid, block, 0, s, p)
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
