// Depthwise 3x3 convolution + per-channel scale + shift + ReLU6, NHWC fp32, gfx950.
//
// Replaces graph nodes DepthwiseConv2dNative -> Mul -> Add -> Relu -> Minimum -> Maximum
// (e.g. #35-39,#44 of the reference's frozen MobileNet; executed by tf_sess.run at
// facerec_test.py:120 / facial_analysis.py:109).
//
// HBM-bound (0.9-2.2 flop/B): the design goal is that every input byte leaves HBM once
// and every load/store instruction moves 1 KiB per wave.
//   * a thread owns ONE float4 of channels at ONE output column and walks DOWN a strip of
//     TH output rows, keeping the 3x3 input window in registers (sliding window: stride 1
//     loads 1 new input row per output row, stride 2 loads 2);
//   * consecutive lanes = consecutive (column, channel-group) float4s, which are contiguous
//     in NHWC, so each of the 3 horizontal taps is one fully coalesced 16 B/lane load; the
//     3x horizontal re-read is served by the CU's L1, not HBM;
//   * vertically adjacent strips (which share 2 halo rows) get workgroup ids on the same
//     XCD (xcd_remap) so the halo re-read hits that XCD's L2.
// Algorithmic bytes per image: 4*C*(H*W + OH*OW) + 36*C + 8*C   (SURVEY 8d).
#include "common.h"

namespace hsefr {

namespace {

struct DwParams {
    const float4* x;
    const float4* w;      // [9][C4] float4 (TF [3,3,C,1] viewed as float4 over C)
    const float4* scale;  // [C4]
    const float4* shift;  // [C4]
    float4* y;
    int H, W, C4, OH, OW, pad_t, pad_l, TH, tiles_h, tiles_x;
    unsigned nwg;
};

__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

template <int STRIDE, int ACT>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(DwParams p) {
    const unsigned bid = xcd_remap(blockIdx.x, p.nwg);
    const int tx = bid % p.tiles_x;
    const int th = (bid / p.tiles_x) % p.tiles_h;
    const int n = bid / (p.tiles_x * p.tiles_h);
    const int t = tx * 256 + threadIdx.x;
    if (t >= p.OW * p.C4) return;
    const int ow = t / p.C4;
    const int c4 = t - ow * p.C4;

    float4 wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = p.w[i * p.C4 + c4];
    const float4 sc = p.scale[c4];
    const float4 sh = p.shift[c4];

    const int iw0 = ow * STRIDE - p.pad_l;  // column of the left tap
    const bool okl = iw0 >= 0, okm = (iw0 + 1 >= 0) && (iw0 + 1 < p.W), okr = iw0 + 2 < p.W;
    const float4* xin = p.x + (size_t)n * p.H * p.W * p.C4 + c4;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_row = [&](int ih, float4* r) {
        if (ih >= 0 && ih < p.H) {
            const float4* row = xin + ((long long)ih * p.W + iw0) * p.C4;
            r[0] = okl ? row[0] : zero;
            r[1] = okm ? row[p.C4] : zero;
            r[2] = okr ? row[2 * p.C4] : zero;
        } else {
            r[0] = r[1] = r[2] = zero;
        }
    };

    const int oh0 = th * p.TH;
    const int oh1 = min(oh0 + p.TH, p.OH);
    float4* yout = p.y + ((size_t)n * p.OH * p.OW + ow) * p.C4 + c4;

    float4 r0[3], r1[3], r2[3];
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
        float4 acc = make_float4(r0[0].x * wk[0].x, r0[0].y * wk[0].y, r0[0].z * wk[0].z, r0[0].w * wk[0].w);
        acc = fma4(r0[1], wk[1], acc);
        acc = fma4(r0[2], wk[2], acc);
        acc = fma4(r1[0], wk[3], acc);
        acc = fma4(r1[1], wk[4], acc);
        acc = fma4(r1[2], wk[5], acc);
        acc = fma4(r2[0], wk[6], acc);
        acc = fma4(r2[1], wk[7], acc);
        acc = fma4(r2[2], wk[8], acc);
        float4 o = fma4(acc, sc, sh);
        o.x = apply_act<ACT>(o.x);
        o.y = apply_act<ACT>(o.y);
        o.z = apply_act<ACT>(o.z);
        o.w = apply_act<ACT>(o.w);
        yout[(size_t)oh * p.OW * p.C4] = o;
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

int launch_dwconv3x3(const float* x, const float* wgt, const float* scale, const float* shift, float* y,
                     int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh, int ow, int act,
                     hipStream_t s) {
    HSEFR_REQUIRE(c % 4 == 0, HSEFR_ERR_UNSUPPORTED, "dwconv3x3: c=%d must be a multiple of 4", c);
    HSEFR_REQUIRE(stride == 1 || stride == 2, HSEFR_ERR_UNSUPPORTED, "dwconv3x3: stride %d", stride);
    HSEFR_REQUIRE(n >= 0 && h > 0 && w > 0 && oh > 0 && ow > 0, HSEFR_ERR_INVALID, "dwconv3x3: bad shape");
    if (n == 0) return HSEFR_OK;
    DwParams p;
    p.x = (const float4*)x; p.w = (const float4*)wgt; p.scale = (const float4*)scale;
    p.shift = (const float4*)shift; p.y = (float4*)y;
    p.H = h; p.W = w; p.C4 = c / 4; p.OH = oh; p.OW = ow; p.pad_t = pad_t; p.pad_l = pad_l;
    p.tiles_x = (ow * p.C4 + 255) / 256;
    // Strip height: as tall as possible (halo re-read = 2/TH of the input for stride 1)
    // while keeping >= ~4 workgroups per CU in flight.
    int th = oh;
    while (th > 4 && (long long)n * p.tiles_x * ((oh + th - 1) / th) < 1024) th = (th + 1) / 2;
    p.TH = th;
    p.tiles_h = (oh + th - 1) / th;
    const long long nwg = (long long)n * p.tiles_x * p.tiles_h;
    HSEFR_REQUIRE(nwg < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "dwconv3x3: grid too large");
    p.nwg = (unsigned)nwg;
    dim3 grid((unsigned)nwg), block(256);
#define HSEFR_DW_LAUNCH(S, A) hipLaunchKernelGGL((dwconv3x3_kernel<S, A>), grid, block, 0, s, p)
    if (stride == 1) {
        if (act == HSEFR_ACT_RELU6) HSEFR_DW_LAUNCH(1, HSEFR_ACT_RELU6);
        else if (act == HSEFR_ACT_RELU) HSEFR_DW_LAUNCH(1, HSEFR_ACT_RELU);
        else if (act == HSEFR_ACT_NONE) HSEFR_DW_LAUNCH(1, HSEFR_ACT_NONE);
        else { set_error("dwconv3x3: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
    } else {
        if (act == HSEFR_ACT_RELU6) HSEFR_DW_LAUNCH(2, HSEFR_ACT_RELU6);
        else if (act == HSEFR_ACT_RELU) HSEFR_DW_LAUNCH(2, HSEFR_ACT_RELU);
        else if (act == HSEFR_ACT_NONE) HSEFR_DW_LAUNCH(2, HSEFR_ACT_NONE);
        else { set_error("dwconv3x3: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
    }
#undef HSEFR_DW_LAUNCH
    return launch_status("dwconv3x3");
}

}  // namespace hsefr
