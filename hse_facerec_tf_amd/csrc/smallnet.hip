// Generic small-CNN kernels for the MTCNN face-detection cascade (facial_analysis.py:334-352, 478-604; mtcnn.pb:
// P-Net 3->10->16->32 channels, R-Net 28/48/64 + FC 128, O-Net 32/64/64/128 + FC 256; 495 832 parameters).
// Channel counts are 10, 16, 28, 48 ... -- not MFMA shapes, and the whole cascade is a few MFLOP per face, so
// these are plain NHWC fp32 VALU kernels: one thread per (output pixel, output channel), output channels fastest
// (weight reads coalesce, the input pixel is a broadcast).  Fully-connected layers run as a VALID convolution
// whose kernel covers the whole feature map.  PReLU is fused: y = v > 0 ? v : alpha[c] * v (the graph spells it
// Relu(v) + alpha * -Relu(-v), nodes pnet/PReLU1/*).
#include <cstdint>

#include "common.h"

namespace hsefr {

namespace {

// Thread = CO consecutive output channels of one output pixel (CO = 4 | 2 | 1, whichever divides Cout): the input value is one
// (broadcast) load for CO FMAs and the CO weights one 16- / 8-byte load.  Every output's FMA chain runs kh, kw, c ascending
// whatever CO is: the same bits as the one-channel form of rounds 1-2 (R-Net / O-Net on a few hundred crops spent 50-80 us per
// layer in it: two loads per FMA, four times the threads).
template <int CO>
__global__ __launch_bounds__(256) void conv2d_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ alpha,
                                                            float* __restrict__ y, int H, int W, int C, int OH, int OW, int Cout,
                                                            int KH, int KW, int stride, int pad_t, int pad_l, long long total, int xvec) {
    typedef float vco __attribute__((ext_vector_type(CO == 1 ? 2 : CO)));      // (CO == 1 reads scalars; the type is unused then)
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;              // index over pixels x (Cout / CO)
    if (i >= total) return;
    const int cg = Cout / CO;
    const int co = (int)(i % cg) * CO;
    const long long p = i / cg;
    const int ow = (int)(p % OW);
    const long long t = p / OW;
    const int oh = (int)(t % OH);
    const long long n = t / OH;
    float acc[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) acc[j] = bias ? bias[co + j] : 0.f;
    const float* xn = x + n * (long long)H * W * C;
    auto wld = [&](const float* q, float* out) __attribute__((always_inline)) {
        if constexpr (CO == 1) out[0] = q[0];
        else {
            const vco v = *(const vco*)q;
#pragma unroll
            for (int j = 0; j < CO; ++j) out[j] = v[j];
        }
    };
    for (int kh = 0; kh < KH; ++kh) {
        const int ih = oh * stride - pad_t + kh;
        if (ih < 0 || ih >= H) continue;
        for (int kw = 0; kw < KW; ++kw) {
            const int iw = ow * stride - pad_l + kw;
            if (iw < 0 || iw >= W) continue;
            const float* xp = xn + ((long long)ih * W + iw) * C;
            const float* wp = w + (long long)((kh * KW + kw) * C) * Cout + co;
            // (c ascending; the input values of four / two channels in one load, the weights of the group in flight together)
            int c = 0;
            if (xvec >= 4 && (C & 3) == 0) {        // xvec: what the ALIGNMENT of x allows (launcher), wave-uniform
                for (; c < C; c += 4) {
                    const float4 xv = *(const float4*)(xp + c);
                    float w0[CO], w1[CO], w2[CO], w3[CO];
                    wld(wp + (long long)c * Cout, w0); wld(wp + (long long)(c + 1) * Cout, w1);
                    wld(wp + (long long)(c + 2) * Cout, w2); wld(wp + (long long)(c + 3) * Cout, w3);
#pragma unroll
                    for (int j = 0; j < CO; ++j) acc[j] = fmaf(xv.w, w3[j], fmaf(xv.z, w2[j], fmaf(xv.y, w1[j], fmaf(xv.x, w0[j], acc[j]))));
                }
            } else if (xvec >= 2 && (C & 1) == 0) {
                for (; c < C; c += 2) {
                    const float2 xv = *(const float2*)(xp + c);
                    float w0[CO], w1[CO];
                    wld(wp + (long long)c * Cout, w0); wld(wp + (long long)(c + 1) * Cout, w1);
#pragma unroll
                    for (int j = 0; j < CO; ++j) acc[j] = fmaf(xv.y, w1[j], fmaf(xv.x, w0[j], acc[j]));
                }
            }
            for (; c < C; ++c) {
                float w0[CO];
                wld(wp + (long long)c * Cout, w0);
#pragma unroll
                for (int j = 0; j < CO; ++j) acc[j] = fmaf(xp[c], w0[j], acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CO; ++j) {
        float a = acc[j];
        if (alpha) a = a > 0.f ? a : alpha[co + j] * a;
        y[p * Cout + co + j] = a;
    }
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
    HSEFR_LAUNCH(conv2d_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, scale, shift, res, y, h, wd, c, oh, ow,
                       cout, kh, kw, stride, pad_t, pad_l, act, total);
    return launch_status("conv2d_f32");
}

int launch_conv2d_direct(const float* x, const float* w, const float* bias, const float* alpha, float* y, int n, int h, int wd, int c,
                         int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && h > 0 && wd > 0 && c > 0 && oh > 0 && ow > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0,
                  HSEFR_ERR_INVALID, "conv2d_direct: bad shape");
    if (n == 0) return HSEFR_OK;
    HSEFR_REQUIRE((long long)n * oh * ow * cout < (1ll << 39), HSEFR_ERR_UNSUPPORTED, "conv2d_direct: too large");
    // output channels per thread: the widest vector the weight rows allow (rows of `cout` floats from a 16-byte aligned base)
    const int co = (cout % 4 == 0 && ((uintptr_t)w & 15) == 0) ? 4 : ((cout % 2 == 0 && ((uintptr_t)w & 7) == 0) ? 2 : 1);
    const long long total = (long long)n * oh * ow * (cout / co);
    // the input side loads four / two channels of a pixel at once: only from a base that is aligned for it (the ABI promises
    // nothing about x; a pixel row is c floats, so c % 4 == 0 keeps every pixel as aligned as the base)
    const int xvec = ((uintptr_t)x & 15) == 0 ? 4 : (((uintptr_t)x & 7) == 0 ? 2 : 1);
#define HSEFR_CONV_DIRECT(CO)                                                                                                          \
    HSEFR_LAUNCH(conv2d_direct_kernel<CO>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, bias, alpha, y, h, wd, c, \
                       oh, ow, cout, kh, kw, stride, pad_t, pad_l, total, xvec)
    if (co == 4) HSEFR_CONV_DIRECT(4);
    else if (co == 2) HSEFR_CONV_DIRECT(2);
    else HSEFR_CONV_DIRECT(1);
#undef HSEFR_CONV_DIRECT
    return launch_status("conv2d_direct");
}

int launch_maxpool_f32(const float* x, float* y, int n, int h, int w, int c, int oh, int ow, int k, int stride, int pad_t, int pad_l,
                       hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0 && oh > 0 && ow > 0 && k > 0 && stride > 0, HSEFR_ERR_INVALID, "maxpool: bad shape");
    if (n == 0) return HSEFR_OK;
    const long long total = (long long)n * oh * ow * c;
    HSEFR_REQUIRE(total < (1ll << 39), HSEFR_ERR_UNSUPPORTED, "maxpool: too large");
    HSEFR_LAUNCH(maxpool_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, h, w, c, oh, ow, k, stride,
                       pad_t, pad_l, total);
    return launch_status("maxpool_f32");
}

}  // namespace hsefr
