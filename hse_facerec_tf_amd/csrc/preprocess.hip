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
// H2D-inclusive rate 12 % under the engine's).  A workgroup owns a band of TB output rows of one image:
//   1. the input rows the band needs are one contiguous byte range: `buffer_load_dwordx4 ... lds` pieces of 1 KiB, all in flight at
//      once (the first version's load -> wait -> ds_write loop paid one memory latency per 1 KiB), the raw-buffer bound returning
//      zeros past the tensor; the column / row tables ([.][8] ints: first tap, up to seven weights, zero-padded as Pillow's are) go
//      to LDS under the DMA;
//   2. horizontal pass, LDS -> LDS: a wave takes input rows, a lane an output pixel whose K weights stay in registers over the
//      rows; the 3K source bytes come as aligned dwords + v_alignbit (not 3K ds_read_u8), the products are v_mad_i32_i24 (a 32-bit
//      v_mul_lo is quarter rate), and the three result bytes of four neighbouring lanes are packed through a quad DPP move into
//      three ds_write_b32;
//   3. vertical pass: elementwise over a row, so a lane takes FOUR consecutive bytes (one ds_read_b32 per tap row) and leaves with
//      one coalesced dword store; the row's weights are wave-uniform.
// Same integer arithmetic, same two roundings to uint8: bit-exact with the two-kernel path and with Pillow.  K = table width
// (3 when enlarging, 5 up to 2x reduction, 7 up to 3x).
// (the clamp goes through an opaque v_med3_i32: left to itself hipcc fuses  sat_u8(a >> 22) | sat_u8(b >> 22) << 8  into gfx950's
// v_ashr_pk_u8_i32, which writes only 16 bits of its destination, and then ORs the other two bytes into an upper half it takes
// for zero -- tools/ashr_pk_probe.hip, profiles/r03_ashr_pk_probe.txt; tools/isa_lint.py refuses the instruction)
__device__ __forceinline__ int u8_round(int acc) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(acc >> 22), "v"(255));
    return r;
}

template <int K>
__global__ __launch_bounds__(256) void pil_resize_u8_band_kernel(const unsigned char* __restrict__ in, unsigned* __restrict__ out,
                                                                 const int* __restrict__ xmin, const int* __restrict__ xcoef, int xk,
                                                                 const int* __restrict__ ymin, const int* __restrict__ ycnt,
                                                                 const int* __restrict__ ycoef, int yk, int H, int W, int oh, int ow,
                                                                 int TB, int cap, int bands, long long in_bytes, int lin_bytes) {
    constexpr int NA = (3 * K + 3) / 4;                                     // dwords holding one pixel's 3K source bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x / bands, band = blockIdx.x - n * bands;
    const int y0 = band * TB, y1 = min(y0 + TB, oh);
    const int r_lo = ymin[y0];
    const int nr = min(ymin[y1 - 1] + ycnt[y1 - 1] - r_lo, cap);           // (the launcher sized `cap` from the scale: never binding)
    const int rowb = W * 3, orow = ow * 3, D = orow >> 2;
    const long long g0 = ((long long)n * H + r_lo) * rowb, g0a = g0 & ~15ll;
    const int phase = (int)(g0 - g0a);
    unsigned char* tmp = lds + lin_bytes;                                   // [nr][orow] horizontally resampled rows
    int* xt = (int*)(tmp + (((size_t)cap * orow + 15) & ~(size_t)15));      // [ow][8]
    int* yt = xt + ow * 8;                                                  // [TB][8]
    {
        // (+3: the tensor's last dword may be cut short; the bytes behind it sit in the same aligned word, are never faulting, and
        // only ever meet zero weights)
        const __amdgpu_buffer_rsrc_t r = make_rsrc(in + g0a, ((in_bytes + 3) & ~3ll) - g0a);
        const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds;
        const int npiece = (phase + nr * rowb + 1023) >> 10;
        for (int p = wave; p < npiece; p += 4)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds0 + p * 1024),
                         "v"(p * 1024 + lane * 16), "s"(r)
                         : "memory", "m0");
    }
    for (int e0 = 0; e0 < ow * 8; e0 += 256 * 8) {                           // eight independent loads per thread, then the writes
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = min(e0 + u * 256 + tid, ow * 8 - 1), xx = e >> 3, f = e & 7;
            v[u] = f == 0 ? xmin[xx] : xcoef[(long long)xx * xk + min(f - 1, xk - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256 + tid, f = e & 7;
            if (e < ow * 8) xt[e] = f == 0 ? v[u] * 3 : (f - 1 < xk ? v[u] : 0);
        }
    }
    if (tid < TB * 8) {
        const int y = min(y0 + (tid >> 3), oh - 1), f = tid & 7;
        yt[tid] = f == 0 ? ymin[y] - r_lo : (f - 1 < yk ? ycoef[(long long)y * yk + f - 1] : 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int xx0 = 0; xx0 < ow; xx0 += 64) {
        const int xx = xx0 + lane;
        const bool on = xx < ow;
        const int4 t0 = *(const int4*)(xt + 8 * (on ? xx : ow - 1)), t1 = *(const int4*)(xt + 8 * (on ? xx : ow - 1) + 4);
        const int w[7] = {t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        const int j = lane & 3;
        for (int r = wave; r < nr; r += 4) {
            const int a = phase + r * rowb + t0.x;
            const unsigned* q = (const unsigned*)(lds + (a & ~3));
            const int sh = (a & 3) * 8;
            unsigned d[NA + 1], s[NA];
#pragma unroll
            for (int i = 0; i <= NA; ++i) d[i] = q[i];
#pragma unroll
            for (int i = 0; i < NA; ++i) s[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], sh);
            int acc[3] = {1 << 21, 1 << 21, 1 << 21};
#pragma unroll
            for (int t = 0; t < K; ++t)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int b = 3 * t + c;
                    acc[c] += __mul24((int)((s[b >> 2] >> (8 * (b & 3))) & 255u), w[t]);
                }
            const unsigned p = (unsigned)u8_round(acc[0]) | ((unsigned)u8_round(acc[1]) << 8) | ((unsigned)u8_round(acc[2]) << 16);
            const unsigned pn = (unsigned)__builtin_amdgcn_mov_dpp((int)p, 0xF9, 0xF, 0xF, true);   // quad_perm [1,2,3,3]: the next lane's pixel
            // four pixels = twelve bytes = three dwords: lanes 0..2 of the quad write one each
            const unsigned dw = (p >> (8 * j)) | (pn << (24 - 8 * j));
            if (on && j < 3) ((unsigned*)(tmp + r * orow))[3 * (xx >> 2) + j] = dw;
        }
    }
    __syncthreads();
    const int nch = (D + 63) >> 6, rows = y1 - y0;
    for (int c = wave; c < rows * nch; c += 4) {
        const int yy = c / nch, d = (c - yy * nch) * 64 + lane;
        // (wave-uniform: through readfirstlane the row offsets and weights are scalar operands, not per-lane arithmetic)
        int w[7];
#pragma unroll
        for (int t = 0; t < 7; ++t) w[t] = t < K ? __builtin_amdgcn_readfirstlane(yt[8 * yy + 1 + t]) : 0;
        const int rb = __builtin_amdgcn_readfirstlane(yt[8 * yy]);
        if (d < D) {
            int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21, a3 = 1 << 21;
            // all K table entries (zero beyond the row's tap count), rows clamped into the band: a uniform trip count
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const unsigned u = ((const unsigned*)(tmp + min(rb + t, nr - 1) * orow))[d];
                a0 += __mul24((int)(u & 255u), w[t]);
                a1 += __mul24((int)((u >> 8) & 255u), w[t]);
                a2 += __mul24((int)((u >> 16) & 255u), w[t]);
                a3 += __mul24((int)(u >> 24), w[t]);
            }
            out[((long long)n * oh + y0 + yy) * D + d] = (unsigned)u8_round(a0) | ((unsigned)u8_round(a1) << 8) |
                                                         ((unsigned)u8_round(a2) << 16) | ((unsigned)u8_round(a3) << 24);
        }
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
    if (mode == 3 && ow % 4 == 0 && xk <= 7 && yk <= 7) {
        // the fused byte-output kernel.  Pillow's tables: first tap = (int)(centre - support + 0.5), one past the last =
        // (int)(centre + support + 0.5) -> a band of TB output rows spans at most (TB - 1) * scale + 2 * support + 1 input rows
        constexpr int TB = 16;
        const double scale = (double)H / oh, support = scale > 1.0 ? scale : 1.0;
        const int cap = (int)((TB - 1) * scale + 2.0 * support) + 2;
        const int kk = xk > yk ? xk : yk;
        const int na = (3 * (kk <= 3 ? 3 : (kk <= 5 ? 5 : 7)) + 3) / 4;
        // input window: 16-byte phase + whole 1 KiB DMA pieces + the over-read of the last pixel's aligned dwords
        const size_t lin_bytes = (((size_t)15 + (size_t)cap * W * 3 + 4 * (na + 1)) + 1023) & ~(size_t)1023;
        const size_t lds_bytes = lin_bytes + (((size_t)cap * ow * 3 + 15) & ~(size_t)15) + (size_t)ow * 32 + TB * 32;
        const int bands = (oh + TB - 1) / TB;
        if (lds_bytes <= 64 * 1024 && (long long)n * bands < (1ll << 31)) {
#define HSEFR_BAND(K)                                                                                                                 \
    HSEFR_LAUNCH(pil_resize_u8_band_kernel<K>, dim3((unsigned)(n * bands)), dim3(256), lds_bytes, s, in, (unsigned*)out, xmin, \
                       xcoef, xk, ymin, ycnt, ycoef, yk, H, W, oh, ow, TB, cap, bands, (long long)n * H * W * 3, (int)lin_bytes)
            if (kk <= 3) HSEFR_BAND(3);
            else if (kk <= 5) HSEFR_BAND(5);
            else HSEFR_BAND(7);
#undef HSEFR_BAND
            return launch_status("pil_resize_fused");
        }
    }
    HSEFR_LAUNCH(pil_resample_h_kernel, dim3(blocks_for(t1)), dim3(256), 0, s, in, tmp, xmin, xcnt, xcoef, xk, H, W, ow, t1);
    HSEFR_LAUNCH(pil_resample_v_kernel, dim3(blocks_for(t2)), dim3(256), 0, s, tmp, out, ymin, ycnt, ycoef, yk, H, oh, ow, t2,
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
        HSEFR_LAUNCH(u8_to_input_kernel, dim3(blocks_for(t)), dim3(256), 0, s, in, out, t, mode, (float)mean[0], (float)mean[1],
                           (float)mean[2], mean[0], mean[1], mean[2]);
    else
        HSEFR_LAUNCH(cv_resize_linear_kernel, dim3(blocks_for(t)), dim3(256), 0, s, in, out, x0, x1, wx1, y0, y1, wy1, H, W, oh,
                           ow, t, mode, (float)mean[0], (float)mean[1], (float)mean[2], mean[0], mean[1], mean[2]);
    return launch_status("cv_resize");
}

}  // namespace hsefr
