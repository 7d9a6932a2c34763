// cv2.resize(..., interpolation=cv2.INTER_AREA) on the device, for the two places the MTCNN cascade of the reference uses
// it (facial_analysis.py:507 -- the image pyramid --, :546 and :575 -- the 24x24 / 48x48 crops fed to R-Net / O-Net), with
// the normalisation (v - 127.5) * 0.0078125 and the (W, H) transposition the nets are fed with fused in.
//
// Host restatement: hse_facerec_tf_amd/preprocess.py resize_area (tables of _area_weights / _area_linear_taps); the
// same coverage weights are derived here per output coordinate instead of being tabulated.
//   pyramid level : uint8 frame -> uint8-rounded level (OpenCV's 8-bit path: float32 accumulation, horizontal taps first,
//                   then vertical, first tap assigns; 2x2 and integer-factor box special cases; bilinear fixed point
//                   when enlarging) -> normalised float32, written transposed [W', H', 3];
//   crops         : per box, the box-sized tile of the frame (zero outside the frame) resized to S x S in float64 (the
//                   reference resizes a float64 tile), normalised, written transposed [n, S, S, 3].
// Arithmetic is kept un-contracted (explicit mul / add) so that the order of roundings is the restatement's.
#include <cmath>

#include "common.h"

namespace hsefr {

namespace {

struct AreaTap { int first, count; float w_first, w_mid, w_last; };      // taps first .. first+count-1 with those weights

// OpenCV's decimation cell of destination index d (preprocess.py _area_weights), ssize >= dsize
__device__ __forceinline__ void area_cell(int d, int ssize, double scale, int& a, int& b, double& cell, bool& has_lo, double& w_lo, bool& has_hi,
                                          double& w_hi) {
    // explicit mul / add: contracted into an fma, `hi` lands one ulp away and with it floor(hi) on integer boundaries
    const double lo = __dmul_rn((double)d, scale), hi = __dadd_rn(lo, scale);
    cell = fmin(scale, (double)ssize - lo);
    a = (int)ceil(lo);
    b = (int)fmin(floor(hi), (double)(ssize - 1));
    a = a < b ? a : b;
    has_lo = (double)a - lo > 1e-3;
    w_lo = ((double)a - lo) / cell;
    has_hi = hi - (double)b > 1e-3;
    w_hi = fmin(fmin(hi - (double)b, 1.0), cell) / cell;
}

// INTER_AREA when enlarging: bilinear taps with the area-mode coordinate rule (preprocess.py _area_linear_taps)
__device__ __forceinline__ void linear_tap(int d, int ssize, double scale, int& s0, int& s1, float& f) {
    const double inv = 1.0 / scale;
    s0 = (int)floor(__dmul_rn((double)d, scale));
    float ff = (float)__dsub_rn((double)(d + 1), __dmul_rn((double)(s0 + 1), inv));
    ff = ff <= 0.f ? 0.f : ff - floorf(ff);
    if (s0 >= ssize - 1) { s0 = ssize - 1; ff = 0.f; }
    s1 = s0 + 1 < ssize ? s0 + 1 : ssize - 1;
    f = ff;
}

// ---- pyramid level: uint8 [H,W,3] -> float32 [dw,dh,3] (transposed), value = (u8(resized) - 127.5) * 0.0078125 ---------------
__global__ __launch_bounds__(256) void area_level_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int sh, int sw, int dh,
                                                         int dw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= dh * dw) return;
    const int dy = i / dw, dx = i - dy * dw;
    const double fx = (double)sw / dw, fy = (double)sh / dh;
    float out[3];
    if (sh == dh && sw == dw) {
        for (int c = 0; c < 3; ++c) out[c] = (float)src[(dy * sw + dx) * 3 + c];
    } else if (fx >= 1.0 && fy >= 1.0) {
        const int ix = (int)rint(fx), iy = (int)rint(fy);
        const double eps = 2.220446049250313e-16;
        if (fabs(fx - ix) < eps && fabs(fy - iy) < eps) {                 // exact integer factors: plain box filter
            for (int c = 0; c < 3; ++c) {
                int tot = 0;
                for (int y = 0; y < iy; ++y)
                    for (int x = 0; x < ix; ++x) tot += src[((dy * iy + y) * sw + dx * ix + x) * 3 + c];
                if (ix == 2 && iy == 2) out[c] = (float)((tot + 2) >> 2);
                else out[c] = fminf(fmaxf(rintf(__fmul_rn((float)tot, (float)(1.0 / (ix * iy)))), 0.f), 255.f);
            }
        } else {
            int xa, xb, ya, yb;
            double xc, yc, xwl, xwh, ywl, ywh;
            bool xhl, xhh, yhl, yhh;
            area_cell(dx, sw, fx, xa, xb, xc, xhl, xwl, xhh, xwh);
            area_cell(dy, sh, fy, ya, yb, yc, yhl, ywl, yhh, ywh);
            const float xm = (float)(1.0 / xc), ym = (float)(1.0 / yc);
            // horizontal pass of source row r: taps in table order, float32, mul then add
            auto hrow = [&](int r, int c) {
                float acc = 0.f;
                const unsigned char* row = src + (long long)r * sw * 3 + c;
                if (xhl) acc = __fadd_rn(acc, __fmul_rn((float)row[(xa - 1) * 3], (float)xwl));
                for (int x = xa; x < xb; ++x) acc = __fadd_rn(acc, __fmul_rn((float)row[x * 3], xm));
                if (xhh) acc = __fadd_rn(acc, __fmul_rn((float)row[xb * 3], (float)xwh));
                return acc;
            };
            for (int c = 0; c < 3; ++c) {
                float acc = 0.f;
                bool first = true;                                          // vertical pass: first tap assigns, later taps add
                auto vadd = [&](int r, float w) {
                    const float term = __fmul_rn(hrow(r, c), w);
                    acc = first ? term : __fadd_rn(acc, term);
                    first = false;
                };
                if (yhl) vadd(ya - 1, (float)ywl);
                for (int y = ya; y < yb; ++y) vadd(y, ym);
                if (yhh) vadd(yb, (float)ywh);
                out[c] = fminf(fmaxf(rintf(acc), 0.f), 255.f);
            }
        }
    } else {                                                                  // enlarging (either axis): OpenCV's 8-bit bilinear fixed point
        int x0, x1, y0, y1;
        float ax, ay;
        linear_tap(dx, sw, fx, x0, x1, ax);
        linear_tap(dy, sh, fy, y0, y1, ay);
        const long long a1 = (long long)rintf(__fmul_rn(ax, 2048.f)), a0 = (long long)rintf(__fmul_rn(__fsub_rn(1.f, ax), 2048.f));
        const long long b1 = (long long)rintf(__fmul_rn(ay, 2048.f)), b0 = (long long)rintf(__fmul_rn(__fsub_rn(1.f, ay), 2048.f));
        for (int c = 0; c < 3; ++c) {
            const long long t0 = src[(y0 * sw + x0) * 3 + c], t1 = src[(y0 * sw + x1) * 3 + c];
            const long long u0 = src[(y1 * sw + x0) * 3 + c], u1 = src[(y1 * sw + x1) * 3 + c];
            const long long h0 = t0 * a0 + t1 * a1, h1 = u0 * a0 + u1 * a1;
            long long v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            out[c] = (float)v;
        }
    }
    float* o = dst + ((long long)dx * dh + dy) * 3;                          // transposed: the nets see (W, H)
    for (int c = 0; c < 3; ++c) o[c] = (float)(((double)out[c] - 127.5) * 0.0078125);
}

// The general decimation case of area_level_kernel with the work laid out for the SMALL levels of the pyramid (round 3: a 19 x 14
// level of a 784 x 588 frame gave 266 threads a 42 x 42 x 3 cell each, byte by byte, one channel at a time: 137 us per level, a
// third of process_image's kernel time).  A workgroup owns one destination row: the column cells go to LDS once; the horizontal
// sums of every source row of the row's cell are computed by all threads -- (source row, destination column) items, three
// channels per pass over the bytes -- into LDS; then a thread per destination column adds them up in table order.  The same
// float32 operations in the same order as the per-pixel kernel: identical results (OpenCV's own order, preprocess.py).
struct XCell { int a, b, flags; float wl, wh, m; };

__global__ __launch_bounds__(256) void area_level_rows_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int sh, int sw,
                                                              int dh, int dw, int hoff) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    XCell* xc = (XCell*)lds;
    float* hbuf = (float*)(lds + hoff);                                       // [rows of the cell][dw][3]
    const int dy = blockIdx.x;
    const double fx = (double)sw / dw, fy = (double)sh / dh;
    for (int dx = threadIdx.x; dx < dw; dx += 256) {
        int xa, xb;
        double c, wl, wh;
        bool hl, hh;
        area_cell(dx, sw, fx, xa, xb, c, hl, wl, hh, wh);
        xc[dx] = XCell{xa, xb, (hl ? 1 : 0) | (hh ? 2 : 0), (float)wl, (float)wh, (float)(1.0 / c)};
    }
    int ya, yb;
    double yc, ywl, ywh;
    bool yhl, yhh;
    area_cell(dy, sh, fy, ya, yb, yc, yhl, ywl, yhh, ywh);
    const float ym = (float)(1.0 / yc);
    const int r_first = yhl ? ya - 1 : ya;
    const int nrows = (yb - ya) + (yhl ? 1 : 0) + (yhh ? 1 : 0);
    __syncthreads();
    for (int item = threadIdx.x; item < nrows * dw; item += 256) {
        const int rr = item / dw, dx = item - rr * dw;
        const XCell c = xc[dx];
        const unsigned char* row = src + (long long)(r_first + rr) * sw * 3;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        if (c.flags & 1) {
            const unsigned char* q = row + (c.a - 1) * 3;
            a0 = __fadd_rn(a0, __fmul_rn((float)q[0], c.wl)); a1 = __fadd_rn(a1, __fmul_rn((float)q[1], c.wl)); a2 = __fadd_rn(a2, __fmul_rn((float)q[2], c.wl));
        }
        for (int x = c.a; x < c.b; ++x) {
            const unsigned char* q = row + x * 3;
            a0 = __fadd_rn(a0, __fmul_rn((float)q[0], c.m)); a1 = __fadd_rn(a1, __fmul_rn((float)q[1], c.m)); a2 = __fadd_rn(a2, __fmul_rn((float)q[2], c.m));
        }
        if (c.flags & 2) {
            const unsigned char* q = row + c.b * 3;
            a0 = __fadd_rn(a0, __fmul_rn((float)q[0], c.wh)); a1 = __fadd_rn(a1, __fmul_rn((float)q[1], c.wh)); a2 = __fadd_rn(a2, __fmul_rn((float)q[2], c.wh));
        }
        float* h = hbuf + (long long)item * 3;
        h[0] = a0; h[1] = a1; h[2] = a2;
    }
    __syncthreads();
    for (int dx = threadIdx.x; dx < dw; dx += 256) {
        float* o = dst + ((long long)dx * dh + dy) * 3;                        // transposed: the nets see (W, H)
        for (int c = 0; c < 3; ++c) {
            float acc = 0.f;
            bool first = true;                                                  // vertical pass: first tap assigns, later taps add
            int rr = 0;
            auto vadd = [&](float w) {
                const float term = __fmul_rn(hbuf[((long long)rr * dw + dx) * 3 + c], w);
                acc = first ? term : __fadd_rn(acc, term);
                first = false;
                ++rr;
            };
            if (yhl) vadd((float)ywl);
            for (int y = ya; y < yb; ++y) vadd(ym);
            if (yhh) vadd((float)ywh);
            const float out = fminf(fmaxf(rintf(acc), 0.f), 255.f);
            o[c] = (float)(((double)out - 127.5) * 0.0078125);
        }
    }
}

// ---- crops: box k = {x1, y1, x2, y2 (1-based inclusive window inside the frame), tx1, ty1 (where it lands in the tile), bw, bh} ----
__global__ __launch_bounds__(256) void area_crops_kernel(const unsigned char* __restrict__ src, const int* __restrict__ boxes,
                                                         float* __restrict__ dst, int sh, int sw, int n, int S) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n * S * S) return;
    const int k = (int)(i / (S * S)), rem = (int)(i - (long long)k * S * S), dy = rem / S, dx = rem - dy * S;
    const int* bx = boxes + k * 8;
    const int x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3], tx1 = bx[4], ty1 = bx[5], bw = bx[6], bh = bx[7];
    (void)sh;
    // tile(y, x) of the box-sized zero-padded tile, 0-based
    auto tile = [&](int y, int x, int c) -> double {
        const int fy = y - (ty1 - 1) + (y1 - 1), fx_ = x - (tx1 - 1) + (x1 - 1);
        const bool in = y >= ty1 - 1 && x >= tx1 - 1 && fy <= y2 - 1 && fx_ <= x2 - 1;
        return in ? (double)src[((long long)fy * sw + fx_) * 3 + c] : 0.0;
    };
    const double fx = (double)bw / S, fy = (double)bh / S;
    double out[3];
    if (bw == S && bh == S) {
        for (int c = 0; c < 3; ++c) out[c] = tile(dy, dx, c);
    } else if (fx >= 1.0 && fy >= 1.0) {
        const int ix = (int)rint(fx), iy = (int)rint(fy);
        const double eps = 2.220446049250313e-16;
        if (fabs(fx - ix) < eps && fabs(fy - iy) < eps) {
            for (int c = 0; c < 3; ++c) {
                double tot = 0.0;
                for (int y = 0; y < iy; ++y)
                    for (int x = 0; x < ix; ++x) tot = __dadd_rn(tot, tile(dy * iy + y, dx * ix + x, c));
                out[c] = __dmul_rn(tot, 1.0 / (ix * iy));
            }
        } else {
            int xa, xb, ya, yb;
            double xc, yc, xwl, xwh, ywl, ywh;
            bool xhl, xhh, yhl, yhh;
            area_cell(dx, bw, fx, xa, xb, xc, xhl, xwl, xhh, xwh);
            area_cell(dy, bh, fy, ya, yb, yc, yhl, ywl, yhh, ywh);
            const double xm = 1.0 / xc, ym = 1.0 / yc;
            for (int c = 0; c < 3; ++c) {
                auto hrow = [&](int r) {
                    double acc = 0.0;
                    if (xhl) acc = __dadd_rn(acc, __dmul_rn(tile(r, xa - 1, c), xwl));
                    for (int x = xa; x < xb; ++x) acc = __dadd_rn(acc, __dmul_rn(tile(r, x, c), xm));
                    if (xhh) acc = __dadd_rn(acc, __dmul_rn(tile(r, xb, c), xwh));
                    return acc;
                };
                double acc = 0.0;
                bool first = true;
                auto vadd = [&](int r, double w) {
                    const double term = __dmul_rn(hrow(r), w);
                    acc = first ? term : __dadd_rn(acc, term);
                    first = false;
                };
                if (yhl) vadd(ya - 1, ywl);
                for (int y = ya; y < yb; ++y) vadd(y, ym);
                if (yhh) vadd(yb, ywh);
                out[c] = acc;
            }
        }
    } else {
        int x0, x1_, y0, y1_;
        float ax, ay;
        linear_tap(dx, bw, fx, x0, x1_, ax);
        linear_tap(dy, bh, fy, y0, y1_, ay);
        const double a1 = (double)ax, a0 = (double)__fsub_rn(1.f, ax), b1 = (double)ay, b0 = (double)__fsub_rn(1.f, ay);
        for (int c = 0; c < 3; ++c) {
            const double h0 = __dadd_rn(__dmul_rn(tile(y0, x0, c), a0), __dmul_rn(tile(y0, x1_, c), a1));
            const double h1 = __dadd_rn(__dmul_rn(tile(y1_, x0, c), a0), __dmul_rn(tile(y1_, x1_, c), a1));
            out[c] = __dadd_rn(__dmul_rn(h0, b0), __dmul_rn(h1, b1));
        }
    }
    float* o = dst + (((long long)k * S + dx) * S + dy) * 3;                  // transposed: the nets see (W, H)
    for (int c = 0; c < 3; ++c) o[c] = (float)((out[c] - 127.5) * 0.0078125);
}

}  // namespace

int launch_area_level(const unsigned char* src, float* dst, int sh, int sw, int dh, int dw, hipStream_t s) {
    HSEFR_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0, HSEFR_ERR_INVALID, "area_level: bad shape");
    // the general decimation case (both factors >= 1, not both integers: every level of MTCNN's 0.709 pyramid) row by row
    const double fx = (double)sw / dw, fy = (double)sh / dh;
    if (fx >= 1.0 && fy >= 1.0 && !(sh == dh && sw == dw)) {
        const double eps = 2.220446049250313e-16;
        const bool integer = std::fabs(fx - std::rint(fx)) < eps && std::fabs(fy - std::rint(fy)) < eps;
        const size_t hoff = ((size_t)dw * sizeof(XCell) + 15) & ~(size_t)15;
        const size_t lds = hoff + ((size_t)fy + 3) * (size_t)dw * 3 * sizeof(float);
        if (!integer && lds <= 64 * 1024) {
            HSEFR_LAUNCH(area_level_rows_kernel, dim3(dh), dim3(256), lds, s, src, dst, sh, sw, dh, dw, (int)hoff);
            return launch_status("area_level_rows");
        }
    }
    HSEFR_LAUNCH(area_level_kernel, dim3((dh * dw + 255) / 256), dim3(256), 0, s, src, dst, sh, sw, dh, dw);
    return launch_status("area_level");
}

int launch_area_crops(const unsigned char* src, const int* boxes, float* dst, int sh, int sw, int n, int size, hipStream_t s) {
    HSEFR_REQUIRE(sh > 0 && sw > 0 && n >= 0 && size > 0, HSEFR_ERR_INVALID, "area_crops: bad shape");
    if (n == 0) return HSEFR_OK;
    const long long total = (long long)n * size * size;
    HSEFR_LAUNCH(area_crops_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, boxes, dst, sh, sw, n, size);
    return launch_status("area_crops");
}

}  // namespace hsefr
