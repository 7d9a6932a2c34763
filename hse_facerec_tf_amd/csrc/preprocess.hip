// Device-side image preparation: uint8 RGB frames -> resized, BGR, mean-subtracted fp32
// NHWC network input, bit-exact with the reference's host code:
//   * facerec_test.py:93-106   misc.imresize(img, size, 'bilinear') = PIL's antialiasing BILINEAR on 8-bit
//     images (two separable fixed-point passes, 22-bit coefficients, each pass rounded to uint8), then
//     float64 BGR - mean, fed to a float32 placeholder;
//   * facial_analysis.py:95-107 cv2.resize INTER_LINEAR on 8-bit images (2x2 taps, 11-bit weights, OpenCV's
//     two-stage rounding), then float32 BGR - mean.
// Integer arithmetic end to end, so parity with PIL / the OpenCV restatement is exact, not a tolerance.
// Coefficient tables are computed on the host in double precision exactly as Pillow's precompute_coeffs
// does (hse_facerec_tf_amd/preprocess_device.py) and cached per (input size, output size).
#include "common.h"

namespace hsefr {

namespace {

// Horizontal PIL pass: in [n,H,W,3] u8 -> out [n,H,ow,3] u8.  thread = one output pixel (3 channels).
__global__ __launch_bounds__(256) void pil_resample_h_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                             const int* __restrict__ xmin, const int* __restrict__ cnt,
                                                             const int* __restrict__ coef, int ksize, int H, int W, int ow,
                                                             long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int xx = (int)(i % ow);
    const long long row = i / ow;  // n*H + y
    const unsigned char* src = in + (row * W + xmin[xx]) * 3;
    const int* k = coef + (long long)xx * ksize;
    const int c = cnt[xx];
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;  // 1 << (PRECISION_BITS - 1), PRECISION_BITS = 22
    for (int x = 0; x < c; ++x) {
        const int w = k[x];
        s0 += src[x * 3 + 0] * w;
        s1 += src[x * 3 + 1] * w;
        s2 += src[x * 3 + 2] * w;
    }
    unsigned char* dst = out + i * 3;
    dst[0] = (unsigned char)min(max(s0 >> 22, 0), 255);
    dst[1] = (unsigned char)min(max(s1 >> 22, 0), 255);
    dst[2] = (unsigned char)min(max(s2 >> 22, 0), 255);
}

// mode: 0 = BGR - mean in float64 then cast (facerec_test.py:95-106), 1 = RGB x/127.5 - 1 (:108-110),
//       2 = BGR - mean in float32 (facial_analysis.py:101-107), 3 = no colour step: `out` receives the resized RGB BYTES
//       [n,oh,ow,3] (a quarter of the fp32 tensor) for hsefr_engine_forward_u8, whose first kernel folds conversion, reversal
//       and mean into its window load
__device__ __forceinline__ void emit(float* dst, int r, int g, int b, int mode, float m0, float m1, float m2, double d0, double d1,
                                     double d2) {
    if (mode == 0) {
        dst[0] = (float)((double)b - d0);
        dst[1] = (float)((double)g - d1);
        dst[2] = (float)((double)r - d2);
    } else if (mode == 1) {
        dst[0] = (float)((double)r / 127.5 - 1.0);
        dst[1] = (float)((double)g / 127.5 - 1.0);
        dst[2] = (float)((double)b / 127.5 - 1.0);
    } else {
        dst[0] = (float)b - m0;
        dst[1] = (float)g - m1;
        dst[2] = (float)r - m2;
    }
}

// Vertical PIL pass + colour handling: in [n,H,ow,3] u8 -> out [n,oh,ow,3] f32.
__global__ __launch_bounds__(256) void pil_resample_v_kernel(const unsigned char* __restrict__ in, float* __restrict__ out,
                                                             const int* __restrict__ ymin, const int* __restrict__ cnt,
                                                             const int* __restrict__ coef, int ksize, int H, int oh, int ow,
                                                             long long total, int mode, float m0, float m1, float m2, double d0,
                                                             double d1, double d2) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int xx = (int)(i % ow);
    const long long t = i / ow;
    const int yy = (int)(t % oh);
    const long long n = t / oh;
    const unsigned char* src = in + ((n * H + ymin[yy]) * ow + xx) * 3;
    const int* k = coef + (long long)yy * ksize;
    const int c = cnt[yy];
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int y = 0; y < c; ++y) {
        const int w = k[y];
        const unsigned char* p = src + (long long)y * ow * 3;
        s0 += p[0] * w;
        s1 += p[1] * w;
        s2 += p[2] * w;
    }
    const int r = min(max(s0 >> 22, 0), 255), g = min(max(s1 >> 22, 0), 255), b = min(max(s2 >> 22, 0), 255);
    if (mode == 3) {       // the resized bytes themselves, RGB: the network's first kernel takes them as they are (stem4_fused.hip)
        unsigned char* d = (unsigned char*)out + i * 3;
        d[0] = (unsigned char)r; d[1] = (unsigned char)g; d[2] = (unsigned char)b;
        return;
    }
    emit(out + i * 3, r, g, b, mode, m0, m1, m2, d0, d1, d2);
}

// cv2.resize INTER_LINEAR (8-bit): taps/weights per axis from the host (OpenCV's float32 coordinate
// arithmetic), out = ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2 with S = horizontal 11-bit sums.
__global__ __launch_bounds__(256) void cv_resize_linear_kernel(const unsigned char* __restrict__ in, float* __restrict__ out,
                                                               const int* __restrict__ x0, const int* __restrict__ x1,
                                                               const int* __restrict__ wx1, const int* __restrict__ y0,
                                                               const int* __restrict__ y1, const int* __restrict__ wy1, int H, int W,
                                                               int oh, int ow, long long total, int mode, float m0, float m1,
                                                               float m2, double d0, double d1, double d2) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int xx = (int)(i % ow);
    const long long t = i / ow;
    const int yy = (int)(t % oh);
    const long long n = t / oh;
    const int a1 = wx1[xx], a0 = 2048 - a1, b1 = wy1[yy], b0 = 2048 - b1;
    const unsigned char* r0 = in + (n * H + y0[yy]) * (long long)W * 3;
    const unsigned char* r1 = in + (n * H + y1[yy]) * (long long)W * 3;
    int v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int s0 = r0[x0[xx] * 3 + c] * a0 + r0[x1[xx] * 3 + c] * a1;
        const int s1 = r1[x0[xx] * 3 + c] * a0 + r1[x1[xx] * 3 + c] * a1;
        v[c] = min(max((((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2, 0), 255);
    }
    if (mode == 3) {
        unsigned char* d = (unsigned char*)out + i * 3;
        d[0] = (unsigned char)v[0]; d[1] = (unsigned char)v[1]; d[2] = (unsigned char)v[2];
        return;
    }
    emit(out + i * 3, v[0], v[1], v[2], mode, m0, m1, m2, d0, d1, d2);
}

// no resize (input already at network size): colour handling only
__global__ __launch_bounds__(256) void u8_to_input_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, long long total,
                                                          int mode, float m0, float m1, float m2, double d0, double d1, double d2) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    if (mode == 3) {
        unsigned char* d = (unsigned char*)out + i * 3;
        d[0] = in[i * 3]; d[1] = in[i * 3 + 1]; d[2] = in[i * 3 + 2];
        return;
    }
    emit(out + i * 3, in[i * 3], in[i * 3 + 1], in[i * 3 + 2], mode, m0, m1, m2, d0, d1, d2);
}

unsigned blocks_for(long long total) { return (unsigned)((total + 255) / 256); }

// Both PIL passes in ONE kernel for the byte-output mode (round 3: the resized bytes feed hsefr_engine_forward_u8, and the two
// per-pixel kernels above -- byte loads and byte stores from global memory, ~170 us per 256 photos -- were what kept the
// H2D-inclusive rate 12 % under the engine's).  A workgroup owns a band of TB output rows of one image: the input rows the
// band needs are one contiguous byte range, staged into LDS with dword loads; the horizontal pass runs LDS -> LDS (bytes: an
// output value mixes input columns with its own weights); the vertical pass is elementwise over a row, so a thread takes FOUR
// consecutive bytes (one ds_read_b32 per tap row) and leaves with one coalesced dword store.  Same integer arithmetic, same two
// roundings to uint8: bit-exact with the two-kernel path and with Pillow.
__global__ __launch_bounds__(256) void pil_resize_u8_fused_kernel(const unsigned char* __restrict__ in, unsigned* __restrict__ out,
                                                                  const int* __restrict__ xmin, const int* __restrict__ xcnt,
                                                                  const int* __restrict__ xcoef, int xk, const int* __restrict__ ymin,
                                                                  const int* __restrict__ ycnt, const int* __restrict__ ycoef, int yk,
                                                                  int H, int W, int oh, int ow, int TB, int cap, int bands,
                                                                  long long in_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / bands, band = blockIdx.x - n * bands;
    const int y0 = band * TB, y1 = min(y0 + TB, oh);
    const int r_lo = ymin[y0];
    const int nr = min(ymin[y1 - 1] + ycnt[y1 - 1] - r_lo, cap);           // (the launcher sized `cap` from the scale: never binding)
    const int rowb = W * 3, orow = ow * 3, D = orow >> 2;
    const long long g0 = ((long long)n * H + r_lo) * rowb;
    const long long g0a = g0 & ~3ll;
    const int phase = (int)(g0 - g0a);
    unsigned char* lin = lds;                                               // [phase + nr * rowb] input bytes
    unsigned char* tmp = lds + (((size_t)cap * rowb + 3 + 15) & ~(size_t)15); // [nr][orow] horizontally resampled rows
    const int nd = (phase + nr * rowb + 3) >> 2;
    for (int i = tid; i < nd; i += 256) {
        const long long a = g0a + 4ll * i;
        unsigned v;
        if (a + 4 <= in_bytes) v = *(const unsigned*)(in + a);
        else {                                                              // the tensor's last dword, cut short
            v = 0;
            for (int b = 0; b < 4; ++b)
                if (a + b < in_bytes) v |= (unsigned)in[a + b] << (8 * b);
        }
        ((unsigned*)lin)[i] = v;
    }
    __syncthreads();
    // the column tables once per workgroup in LDS ([ow][8] ints: byte offset of the first tap, tap count, up to six weights): every
    // item of the horizontal pass would otherwise fetch them from global memory again
    int* xt = (int*)(tmp + (size_t)cap * orow);
    const bool xt_ok = xk <= 6;
    if (xt_ok)
        for (int i = tid; i < ow * 8; i += 256) {
            const int xx = i >> 3, f = i & 7;
            xt[i] = f == 0 ? xmin[xx] * 3 : (f == 1 ? xcnt[xx] : (f - 2 < xk ? xcoef[(long long)xx * xk + f - 2] : 0));
        }
    __syncthreads();
    for (int item = tid; item < nr * ow; item += 256) {
        const int r = item / ow, xx = item - r * ow;
        const unsigned char* src = lin + phase + r * rowb + (xt_ok ? xt[8 * xx] : xmin[xx] * 3);
        const int* k = xt_ok ? xt + 8 * xx + 2 : xcoef + (long long)xx * xk;
        const int c = xt_ok ? xt[8 * xx + 1] : xcnt[xx];
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        for (int x = 0; x < c; ++x) {
            const int w = k[x];
            s0 += src[x * 3 + 0] * w;
            s1 += src[x * 3 + 1] * w;
            s2 += src[x * 3 + 2] * w;
        }
        unsigned char* dst = tmp + r * orow + xx * 3;
        dst[0] = (unsigned char)min(max(s0 >> 22, 0), 255);
        dst[1] = (unsigned char)min(max(s1 >> 22, 0), 255);
        dst[2] = (unsigned char)min(max(s2 >> 22, 0), 255);
    }
    __syncthreads();
    for (int item = tid; item < (y1 - y0) * D; item += 256) {
        const int yy = item / D, d = item - yy * D, y = y0 + yy;
        const int* k = ycoef + (long long)y * yk;
        const int rb = ymin[y] - r_lo;
        int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21, a3 = 1 << 21;
        // all `yk` table entries (zero beyond the row's tap count: Pillow's table is zero-padded), rows clamped into the band: a
        // uniform trip count keeps the loop a plain one
        for (int t = 0; t < yk; ++t) {
            const int w = k[t];
            const unsigned u = ((const unsigned*)(tmp + min(rb + t, nr - 1) * orow))[d];
            a0 += (int)(u & 255u) * w;
            a1 += (int)((u >> 8) & 255u) * w;
            a2 += (int)((u >> 16) & 255u) * w;
            a3 += (int)(u >> 24) * w;
        }
        const unsigned o = (unsigned)min(max(a0 >> 22, 0), 255) | ((unsigned)min(max(a1 >> 22, 0), 255) << 8) |
                           ((unsigned)min(max(a2 >> 22, 0), 255) << 16) | ((unsigned)min(max(a3 >> 22, 0), 255) << 24);
        out[((long long)n * oh + y) * D + d] = o;
    }
}

}  // namespace

int launch_pil_resize(const unsigned char* in, unsigned char* tmp, float* out, int n, int H, int W, int oh, int ow, const int* xmin,
                      const int* xcnt, const int* xcoef, int xk, const int* ymin, const int* ycnt, const int* ycoef, int yk,
                      int mode, const double* mean, hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && H > 0 && W > 0 && oh > 0 && ow > 0, HSEFR_ERR_INVALID, "pil_resize: bad shape");
    HSEFR_REQUIRE(mode >= 0 && mode <= 3, HSEFR_ERR_INVALID, "preprocess: mode %d", mode);
    if (n == 0) return HSEFR_OK;
    const long long t1 = (long long)n * H * ow, t2 = (long long)n * oh * ow;
    HSEFR_REQUIRE(t1 < (1ll << 39) && t2 < (1ll << 39), HSEFR_ERR_UNSUPPORTED, "pil_resize: too large");
    if (mode == 3 && ow % 4 == 0) {
        // the fused byte-output kernel: a band of TB output rows needs at most TB * scale + 2 * support (+ rounding) input rows
        constexpr int TB = 16;
        const double scale = (double)H / oh, support = scale > 1.0 ? scale : 1.0;
        const int cap = (int)(TB * scale + 2.0 * support) + 4;
        const size_t lds_bytes = (((size_t)cap * W * 3 + 3 + 15) & ~(size_t)15) + (size_t)cap * ow * 3 + (size_t)ow * 32;
        const int bands = (oh + TB - 1) / TB;
        if (lds_bytes <= 60 * 1024 && (long long)n * bands < (1ll << 31)) {
            hipLaunchKernelGGL(pil_resize_u8_fused_kernel, dim3((unsigned)(n * bands)), dim3(256), lds_bytes, s, in, (unsigned*)out, xmin, xcnt, xcoef, xk,
                               ymin, ycnt, ycoef, yk, H, W, oh, ow, TB, cap, bands, (long long)n * H * W * 3);
            return launch_status("pil_resize_fused");
        }
    }
    hipLaunchKernelGGL(pil_resample_h_kernel, dim3(blocks_for(t1)), dim3(256), 0, s, in, tmp, xmin, xcnt, xcoef, xk, H, W, ow, t1);
    hipLaunchKernelGGL(pil_resample_v_kernel, dim3(blocks_for(t2)), dim3(256), 0, s, tmp, out, ymin, ycnt, ycoef, yk, H, oh, ow, t2,
                       mode, (float)mean[0], (float)mean[1], (float)mean[2], mean[0], mean[1], mean[2]);
    return launch_status("pil_resize");
}

int launch_cv_resize(const unsigned char* in, float* out, int n, int H, int W, int oh, int ow, const int* x0, const int* x1,
                     const int* wx1, const int* y0, const int* y1, const int* wy1, int mode, const double* mean, hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && H > 0 && W > 0 && oh > 0 && ow > 0, HSEFR_ERR_INVALID, "cv_resize: bad shape");
    HSEFR_REQUIRE(mode >= 0 && mode <= 3, HSEFR_ERR_INVALID, "preprocess: mode %d", mode);
    if (n == 0) return HSEFR_OK;
    const long long t = (long long)n * oh * ow;
    if (H == oh && W == ow)
        hipLaunchKernelGGL(u8_to_input_kernel, dim3(blocks_for(t)), dim3(256), 0, s, in, out, t, mode, (float)mean[0], (float)mean[1],
                           (float)mean[2], mean[0], mean[1], mean[2]);
    else
        hipLaunchKernelGGL(cv_resize_linear_kernel, dim3(blocks_for(t)), dim3(256), 0, s, in, out, x0, x1, wx1, y0, y1, wy1, H, W, oh,
                           ow, t, mode, (float)mean[0], (float)mean[1], (float)mean[2], mean[0], mean[1], mean[2]);
    return launch_status("cv_resize");
}

}  // namespace hsefr
