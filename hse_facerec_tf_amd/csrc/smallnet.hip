// Generic small-CNN kernels for the MTCNN face-detection cascade (facial_analysis.py:334-352, 478-604; mtcnn.pb:
// P-Net 3->10->16->32 channels, R-Net 28/48/64 + FC 128, O-Net 32/64/64/128 + FC 256; 495 832 parameters).
// Channel counts are 10, 16, 28, 48 ... -- not MFMA shapes, and the whole cascade is a few MFLOP per face, so
// these are plain NHWC fp32 VALU kernels: one thread per (output pixel, output channel), output channels fastest
// (weight reads coalesce, the input pixel is a broadcast).  Fully-connected layers run as a VALID convolution
// whose kernel covers the whole feature map.  PReLU is fused: y = v > 0 ? v : alpha[c] * v (the graph spells it
// Relu(v) + alpha * -Relu(-v), nodes pnet/PReLU1/*).
#include "common.h"

namespace hsefr {

namespace {

__global__ __launch_bounds__(256) void conv2d_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ alpha,
                                                            float* __restrict__ y, int H, int W, int C, int OH, int OW, int Cout,
                                                            int KH, int KW, int stride, int pad_t, int pad_l, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % Cout);
    const long long p = i / Cout;
    const int ow = (int)(p % OW);
    const long long t = p / OW;
    const int oh = (int)(t % OH);
    const long long n = t / OH;
    float acc = bias ? bias[co] : 0.f;
    const float* xn = x + n * (long long)H * W * C;
    for (int kh = 0; kh < KH; ++kh) {
        const int ih = oh * stride - pad_t + kh;
        if (ih < 0 || ih >= H) continue;
        for (int kw = 0; kw < KW; ++kw) {
            const int iw = ow * stride - pad_l + kw;
            if (iw < 0 || iw >= W) continue;
            const float* xp = xn + ((long long)ih * W + iw) * C;
            const float* wp = w + (long long)((kh * KW + kw) * C) * Cout + co;
            // (same FMA chain, c ascending; the input values of four / two channels in one load, the weights of the group in flight together)
            int c = 0;
            if ((C & 3) == 0) {
                for (; c < C; c += 4) {
                    const float4 xv = *(const float4*)(xp + c);
                    const float w0 = wp[(long long)c * Cout], w1 = wp[(long long)(c + 1) * Cout], w2 = wp[(long long)(c + 2) * Cout], w3 = wp[(long long)(c + 3) * Cout];
                    acc = fmaf(xv.w, w3, fmaf(xv.z, w2, fmaf(xv.y, w1, fmaf(xv.x, w0, acc))));
                }
            } else if ((C & 1) == 0) {
                for (; c < C; c += 2) {
                    const float2 xv = *(const float2*)(xp + c);
                    const float w0 = wp[(long long)c * Cout], w1 = wp[(long long)(c + 1) * Cout];
                    acc = fmaf(xv.y, w1, fmaf(xv.x, w0, acc));
                }
            }
            for (; c < C; ++c) acc = fmaf(xp[c], wp[(long long)c * Cout], acc);
        }
    }
    if (alpha) acc = acc > 0.f ? acc : alpha[co] * acc;
    y[i] = acc;
}

// max-pool k x k / stride, windows clipped to the image (== TF's -inf padding for SAME)
__global__ __launch_bounds__(256) void maxpool_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C,
                                                          int OH, int OW, int K, int stride, int pad_t, int pad_l, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    const int ow = (int)(p % OW);
    const long long t = p / OW;
    const int oh = (int)(t % OH);
    const long long n = t / OH;
    float m = -INFINITY;
    for (int kh = 0; kh < K; ++kh) {
        const int ih = oh * stride - pad_t + kh;
        if (ih < 0 || ih >= H) continue;
        for (int kw = 0; kw < K; ++kw) {
            const int iw = ow * stride - pad_l + kw;
            if (iw < 0 || iw >= W) continue;
            m = fmaxf(m, x[((n * H + ih) * W + iw) * C + c]);
        }
    }
    y[i] = m;
}

// General NHWC fp32 convolution with the epilogue of the bf16 ResNet kernels (per-channel scale + shift, optional residual,
// activation): the fp32-grade mode of ResNet-style graphs (facerec_test.py:213 at the 1e-4 bar; SURVEY 7 "hard parts").  A
// thread owns four consecutive output channels of one pixel: the input value is a broadcast, the four weights one 16-byte
// load.  Exact fp32 FMA chains -- a correctness mode, an order of magnitude slower than the bf16 MFMA path.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void conv2d_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
                                                         int H, int W, int C, int OH, int OW, int Cout, int KH, int KW, int stride, int pad_t,
                                                         int pad_l, int act, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4n = Cout / 4;
    const int co = (int)(i % c4n) * 4;
    const long long p = i / c4n;
    const int ow = (int)(p % OW);
    const long long t = p / OW;
    const int oh = (int)(t % OH);
    const long long n = t / OH;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* xn = x + n * (long long)H * W * C;
    for (int kh = 0; kh < KH; ++kh) {
        const int ih = oh * stride - pad_t + kh;
        if (ih < 0 || ih >= H) continue;
        for (int kw = 0; kw < KW; ++kw) {
            const int iw = ow * stride - pad_l + kw;
            if (iw < 0 || iw >= W) continue;
            const float* xp = xn + ((long long)ih * W + iw) * C;
            const float* wp = w + (long long)((kh * KW + kw) * C) * Cout + co;
            for (int c = 0; c < C; ++c) {
                const f32x4 wv = *(const f32x4*)(wp + (long long)c * Cout);
                const float xv = xp[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(xv, wv[e], acc[e]);
            }
        }
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = fmaf(acc[e], scale ? scale[co + e] : 1.f, shift ? shift[co + e] : 0.f);
        if (res) v += res[p * Cout + co + e];
        o[e] = apply_act_rt(v, act);
    }
    *(f32x4*)(y + p * Cout + co) = o;
}

}  // namespace

int launch_conv2d_f32(const float* x, const float* w, const float* scale, const float* shift, const float* res, float* y, int n, int h,
                      int wd, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, int act, hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && h > 0 && wd > 0 && c > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && oh > 0 && ow > 0, HSEFR_ERR_INVALID,
                  "conv2d_f32: bad shape");
    HSEFR_REQUIRE(cout % 4 == 0, HSEFR_ERR_UNSUPPORTED, "conv2d_f32: cout=%d must be a multiple of 4", cout);
    if (n == 0) return HSEFR_OK;
    const long long total = (long long)n * oh * ow * (cout / 4);
    HSEFR_REQUIRE((total + 255) / 256 < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv2d_f32: grid too large");
    hipLaunchKernelGGL(conv2d_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, scale, shift, res, y, h, wd, c, oh, ow,
                       cout, kh, kw, stride, pad_t, pad_l, act, total);
    return launch_status("conv2d_f32");
}

int launch_conv2d_direct(const float* x, const float* w, const float* bias, const float* alpha, float* y, int n, int h, int wd, int c,
                         int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && h > 0 && wd > 0 && c > 0 && oh > 0 && ow > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0,
                  HSEFR_ERR_INVALID, "conv2d_direct: bad shape");
    if (n == 0) return HSEFR_OK;
    const long long total = (long long)n * oh * ow * cout;
    HSEFR_REQUIRE(total < (1ll << 39), HSEFR_ERR_UNSUPPORTED, "conv2d_direct: too large");
    hipLaunchKernelGGL(conv2d_direct_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, bias, alpha, y, h, wd, c,
                       oh, ow, cout, kh, kw, stride, pad_t, pad_l, total);
    return launch_status("conv2d_direct");
}

int launch_maxpool_f32(const float* x, float* y, int n, int h, int w, int c, int oh, int ow, int k, int stride, int pad_t, int pad_l,
                       hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0 && oh > 0 && ow > 0 && k > 0 && stride > 0, HSEFR_ERR_INVALID, "maxpool: bad shape");
    if (n == 0) return HSEFR_OK;
    const long long total = (long long)n * oh * ow * c;
    HSEFR_REQUIRE(total < (1ll << 39), HSEFR_ERR_UNSUPPORTED, "maxpool: too large");
    hipLaunchKernelGGL(maxpool_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, h, w, c, oh, ow, k, stride,
                       pad_t, pad_l, total);
    return launch_status("maxpool_f32");
}

}  // namespace hsefr
