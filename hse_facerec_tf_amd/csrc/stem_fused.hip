// The whole MobileNet stem in one kernel: conv1 3x3/2 (3 -> 32, + shift + ReLU6) -> depthwise 3x3/1 (+ scale + shift +
// ReLU6) -> pointwise 1x1 (32 -> 64, + shift + act), NHWC fp32, gfx950.
//
// Replaces graph nodes #30-#49 (Conv2D conv1, Add, Relu, Minimum, Maximum; DepthwiseConv2dNative conv_dw_1, Mul, Add,
// Relu, Minimum, Maximum; Conv2D conv_pw_1, Add, Relu, Minimum, Maximum), run by tf_sess.run at facerec_test.py:120 /
// facial_analysis.py:109.  Unfused, conv1's 96x96x32 output (1.18 MB per face) is written to HBM and read straight
// back by the first block: 604 MB per 256-face batch, 9 % of everything the trunk moves.  Here the image goes in
// (0.44 MB per face) and the 96x96x64 block output comes out (2.36 MB); nothing in between leaves the CU.
//
// One workgroup (256 threads) owns an 8 x 16 patch of the block's output, all 64 channels:
//   A  gather   180 threads fetch the 3x3x3 input window of one conv1 pixel each -- the 10 x 18 region the depthwise
//               needs, halo included -- with 9 dwordx3 loads from clamped addresses (conv padding = 0/1 factors) and
//               write it as one 128-B im2col row (k = dy*9 + dx*3 + ci, zero-padded to 32);
//   B  conv1    [192 x 32] . [32 x 32] on v_mfma_f32_16x16x4_f32 (exact fp32 -- the image is not bounded, so no f16
//               split here); operands swapped so a lane ends up with 4 consecutive channels of one pixel; + shift,
//               ReLU6, and pixels outside the 96 x 96 map forced to 0 (they ARE the depthwise's zero padding) -> LDS;
//   C  depthwise from LDS, sliding 3x3 window down 4 rows per thread, + scale + shift + ReLU6, scaled by 2^12 and split
//               into f16 hi + lo -> the GEMM's A tile in LDS (pwconv_f16s.hip's row format);
//   D  pointwise [128 x 32] . [32 x 64]: 12 v_mfma_f32_32x32x16_f16 per wave (al*bh + ah*bl + ah*bh), weights resident;
//   E  epilogue  acc * descale + shift, activation, transposed through LDS into whole 128-B lines, buffer stores.
// The next patch's gather is in flight during B-E.  Halo recomputation: 180 conv1 pixels per 128 outputs (1.4x of a
// 4 GFLOP layer); HBM reads of the halo hit L2 (patch ids are XCD-remapped).
#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };

struct StemParams {
    const float* x;        // [N,H,W,3]
    const float* cw;       // conv1 kernel, TF HWIO [3,3,3,32]
    const float* cshift;   // [32]
    const float4* wd;      // depthwise [9][8] float4
    const float4* dscale;  // [8]
    const float4* dshift;  // [8]
    const float* wsplit;   // pointwise split rows [64][1][64 f16]
    const float* descale;  // [64]
    const float* pshift;   // [64]
    float* y;              // [N,OH,OW,64]
    int H, W, OH, OW, cpad_t, cpad_l, tiles_w, tiles_h;
    unsigned total;
    float a_scale;
    int reverse;
    unsigned long long* stamps;   // diagnostic builds (-DHSEFR_STEM_STAMPS) only
};



constexpr int TW = 16, TH = 8, RW = TW + 2, RH = TH + 2, RPIX = RW * RH;   // conv1 region 10 x 18 = 180 pixels
constexpr int RROWS = 192;                                                   // padded to 12 MFMA row blocks of 16

__device__ __forceinline__ int swz32(int row, int chunk) { return row * 32 + 4 * (chunk ^ ((row >> 1) & 7) ^ ((row & 1) << 2)); }       // floats
__device__ __forceinline__ int swzb(int row, int chunk) { return row * 128 + 16 * (chunk ^ ((row >> 1) & 7) ^ ((row & 1) << 2)); }      // bytes; b64 writes of adjacent rows land in different halves of the 128-B bank window
__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void stem_fused_kernel(StemParams p) {
    __shared__ __attribute__((aligned(16))) float Ic[RROWS * 32];      // im2col rows; later the epilogue's transpose scratch
    __shared__ __attribute__((aligned(16))) float Co[RROWS * 32];      // conv1 output region [pixel][32 ch]
    __shared__ __attribute__((aligned(16))) float Cw[32 * 32];         // conv1 weights [n][k]
    __shared__ __attribute__((aligned(16))) unsigned char As[128 * 128];   // split-f16 depthwise result (GEMM A tile)
    __shared__ __attribute__((aligned(16))) unsigned char Bw[64 * 128];    // pointwise split rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int l16 = lane & 15, q4 = lane >> 4;

    // ---- weights, once per workgroup ----
    for (int i = tid; i < 32 * 32; i += 256) {
        const int n = i >> 5, k = i & 31;
        Cw[swz32(n, k >> 2) + (k & 3)] = k < 27 ? p.cw[k * 32 + n] : 0.f;
    }
    for (int i = tid; i < 64 * 8; i += 256) {
        const int r = i >> 3, ch = i & 7;
        *(f32x4*)(&Bw[swzb(r, ch)]) = *(const f32x4*)(p.wsplit + r * 32 + ch * 4);
    }
    // depthwise constants of this thread's channel quad (C = 32: fixed for the whole kernel)
    const int dtw = (tid & 127) >> 3, c4l = tid & 7, drow0 = (tid >> 7) * 4;
    float4 wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = p.wd[i * 8 + c4l];
    const float4 dsc = p.dscale[c4l], dsh = p.dshift[c4l];
    // conv1 epilogue constants: lane owns channels n = nb*16 + 4*q4 + (0..3)
    f32x4 csh[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) csh[nb] = *(const f32x4*)(p.cshift + nb * 16 + 4 * q4);
    // pointwise epilogue constants (transposed layout: lane owns channels 4*ech .. +3 of its wave's 32-column half)
    const int wm = wave >> 1, wn = wave & 1;
    const int erow = lane >> 3, ech = lane & 7;
    const f32x4 pds = *(const f32x4*)(p.descale + wn * 32 + 4 * ech), psh = *(const f32x4*)(p.pshift + wn * 32 + 4 * ech);

    // ---- stage A helpers: gather one conv1 pixel's 3x3x3 window (branch-free), scatter it as an im2col row ----
    F3 g[9];
    float mk[9];
    float cvalid = 0.f;   // 1 if this thread's conv1 pixel lies inside the map, 0 if it is depthwise padding
    // Patch cursor.  A workgroup's patches t = blockIdx.x + i*gridDim.x sit gridDim.x/8 apart in the XCD-remapped
    // order (same XCD, next slot), so the (image, patch row, patch column) triple is advanced with carries instead of
    // being re-derived by six integer divisions per patch (they cost ~2k cycles per patch on the scalar unit).
    struct Cur { int n, th, tw; };
    auto decode = [&](unsigned t) {
        const unsigned lt = xcd_remap_dir(t, p.total, p.reverse);
        Cur c;
        c.tw = lt % p.tiles_w;
        c.th = (lt / p.tiles_w) % p.tiles_h;
        c.n = lt / (p.tiles_w * p.tiles_h);
        return c;
    };
    const int stride_lt = gridDim.x / 8;                       // launch guarantees gridDim.x % 8 == 0 when it loops
    const int dtw_ = stride_lt % p.tiles_w, dth_ = (stride_lt / p.tiles_w) % p.tiles_h, dn_ = stride_lt / (p.tiles_w * p.tiles_h);
    auto advance = [&](Cur c) {
        if (!p.reverse) {
            c.tw += dtw_; if (c.tw >= p.tiles_w) { c.tw -= p.tiles_w; c.th += 1; }
            c.th += dth_; if (c.th >= p.tiles_h) { c.th -= p.tiles_h; c.n += 1; }
            c.n += dn_;
        } else {
            c.tw -= dtw_; if (c.tw < 0) { c.tw += p.tiles_w; c.th -= 1; }
            c.th -= dth_; if (c.th < 0) { c.th += p.tiles_h; c.n -= 1; }
            c.n -= dn_;
        }
        return c;
    };
    auto gather = [&](Cur c) {
        const int n = c.n, oh0 = c.th * TH, ow0 = c.tw * TW;
        const int rp = tid < RPIX ? tid : RPIX - 1;          // threads 180..255 shadow the last pixel (rows never used)
        const int cy = oh0 - 1 + rp / RW, cx = ow0 - 1 + rp % RW;    // conv1 output coordinates of this region pixel
        cvalid = (cy >= 0 && cy < p.OH && cx >= 0 && cx < p.OW) ? 1.f : 0.f;
        const int ih0 = cy * 2 - p.cpad_t, iw0 = cx * 2 - p.cpad_l;
        const float* img = p.x + (size_t)n * p.H * p.W * 3;       // uniform
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ih = ih0 + dy, ihc = min(max(ih, 0), p.H - 1);
            const float my = (ih >= 0 && ih < p.H) ? 1.f : 0.f;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int iw = iw0 + dx, iwc = min(max(iw, 0), p.W - 1);
                g[dy * 3 + dx] = *(const F3*)(img + (unsigned)(ihc * p.W + iwc) * 3u);
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
        if (tid < RROWS) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                f32x4 o;
                o[0] = v[4 * c]; o[1] = v[4 * c + 1]; o[2] = v[4 * c + 2]; o[3] = v[4 * c + 3];
                *(f32x4*)(&Ic[swz32(tid, c)]) = o;
            }
        }
    };

    unsigned t = blockIdx.x;
    if (t >= p.total) return;
    Cur cur = decode(t);
    gather(cur);
    scatter();
    float cv = cvalid;       // validity of the pixel whose row this thread scattered (consumed in stage B via LDS below)
    __shared__ float Cv[RROWS];
    if (tid < RROWS) Cv[tid] = tid < RPIX ? cv : 0.f;
    __syncthreads();

    STEM_STAMP_DECL;
    while (true) {
        const unsigned tn = t + gridDim.x;
        const bool more = tn < p.total;
        const int n = cur.n, oh0 = cur.th * TH, ow0 = cur.tw * TW;
        const Cur nxt = advance(cur);
        if (more) gather(nxt);                                 // next patch's window loads fly during stages B-E
        STEM_STAMP(0);

        // ---- stage B: conv1 on the fp32 MFMA, wave w = region rows [48w, 48w + 48) ----
        {
            f32x4 acc[3][2];
#pragma unroll
            for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 a[3], b[2];
#pragma unroll
                for (int mb = 0; mb < 3; ++mb) a[mb] = *(const f32x4*)(&Ic[swz32(wave * 48 + mb * 16 + l16, 4 * s + q4)]);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) b[nb] = *(const f32x4*)(&Cw[swz32(nb * 16 + l16, 4 * s + q4)]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nb][e], a[mb][e], acc[mb][nb], 0, 0, 0);
            }
            // lane: pixel m = 48w + 16mb + l16, channels nb*16 + 4*q4 + (0..3)
#pragma unroll
            for (int mb = 0; mb < 3; ++mb) {
                const int m = wave * 48 + mb * 16 + l16;
                const float valid = Cv[m];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = relu6(acc[mb][nb][e] + csh[nb][e]) * valid;
                    *(f32x4*)(&Co[swz32(m, nb * 4 + q4)]) = o;
                }
            }
        }
        STEM_STAMP(1);
        __syncthreads();     // Co complete; Ic free
        STEM_STAMP(2);

        // ---- stage C: depthwise 3x3/1 from LDS -> split-f16 A tile ----
        {
            auto tap = [&](int ry, int rx) {
                const int r = ry * RW + rx;
                const f32x4 v = *(const f32x4*)(&Co[swz32(r, c4l)]);
                return make_float4(v[0], v[1], v[2], v[3]);
            };
            struct Row { float4 l, m, r; };
            auto load_row = [&](int ry) { Row q; q.l = tap(ry, dtw); q.m = tap(ry, dtw + 1); q.r = tap(ry, dtw + 2); return q; };
            auto row_sum = [&](const Row& q, int b, float4 s) {
                s = fma4(q.l, wk[b], s);
                s = fma4(q.m, wk[b + 1], s);
                return fma4(q.r, wk[b + 2], s);
            };
            Row r0 = load_row(drow0), r1 = load_row(drow0 + 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const Row r2 = load_row(drow0 + j + 2);
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                s = row_sum(r0, 0, s);
                s = row_sum(r1, 3, s);
                s = row_sum(r2, 6, s);
                const float4 o = fma4(s, dsc, dsh);
                f32x4 v;
                v[0] = relu6(o.x); v[1] = relu6(o.y); v[2] = relu6(o.z); v[3] = relu6(o.w);
                v = v * p.a_scale;
                const f16x4 hi = __builtin_convertvector(v, f16x4);
                const f16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), f16x4);
                const int R = (drow0 + j) * TW + dtw;
                *(f16x4*)(&As[swzb(R, c4l >> 1) + 8 * (c4l & 1)]) = hi;
                *(f16x4*)(&As[swzb(R, 4 + (c4l >> 1)) + 8 * (c4l & 1)]) = lo;
                r0 = r1; r1 = r2;
            }
        }
        STEM_STAMP(3);
        __syncthreads();     // A tile complete; Co free
        STEM_STAMP(2);

        // ---- stage D: pointwise on the f16 MFMA (operands swapped: weights first), wave tile 64 rows x 32 channels ----
        f32x16 pacc[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[mi][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 ah[2], al[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                ah[mi] = *(const f16x8*)(&As[swzb(wm * 64 + mi * 32 + li, 2 * s + lh)]);
                al[mi] = *(const f16x8*)(&As[swzb(wm * 64 + mi * 32 + li, 4 + 2 * s + lh)]);
            }
            const f16x8 bh = *(const f16x8*)(&Bw[swzb(wn * 32 + li, 2 * s + lh)]);
            const f16x8 bl = *(const f16x8*)(&Bw[swzb(wn * 32 + li, 4 + 2 * s + lh)]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                pacc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al[mi], pacc[mi], 0, 0, 0);
                pacc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah[mi], pacc[mi], 0, 0, 0);
                pacc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah[mi], pacc[mi], 0, 0, 0);
            }
        }

        STEM_STAMP(4);
        // ---- stage E: epilogue through a wave-private 32 x 128 B scratch in Ic (free since stage B) ----
        {
            float* scr = Ic + wave * 1024;
            // one resource per image; a pixel outside the map gets an offset beyond it and the store is dropped (no branch)
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + (size_t)n * p.OH * p.OW * 64, (long long)p.OH * p.OW * 256);
            const unsigned ycol = (unsigned)(wn * 32 + 4 * ech) * 4u;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = pacc[mi][4 * j + e];
                    *(f32x4*)(scr + li * 32 + 4 * ((2 * j + lh) ^ (li & 7))) = v;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = erow + 8 * i;
                    const f32x4 v = *(const f32x4*)(scr + r * 32 + 4 * (ech ^ (r & 7)));
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = apply_act<ACT>(fmaf(v[e], pds[e], psh[e]));
                    const int R = wm * 64 + mi * 32 + r;              // patch row R = th*16 + tw
                    const int oh = oh0 + (R >> 4), ow = ow0 + (R & 15);
                    const unsigned voff = (oh < p.OH && ow < p.OW) ? (unsigned)(oh * p.OW + ow) * 256u + ycol : 0x80000000u;
                    bstore16(o, ry, voff, 0);
                }
            }
        }
        STEM_STAMP(5);
        STEM_STAMP_COUNT;
        if (!more) break;
        __syncthreads();     // every wave is done with its scratch (Ic) and with As
        STEM_STAMP(2);
        scatter();
        if (tid < RROWS) Cv[tid] = tid < RPIX ? cvalid : 0.f;
        STEM_STAMP(6);
        __syncthreads();
        STEM_STAMP(2);
        t = tn;
        cur = nxt;
    }
    STEM_STAMP_FLUSH(p.stamps, lane, wave);
}

}  // namespace

bool stem_fused_supported(int cin, int cmid, int cout, int conv_stride, int dw_stride, int kh, int kw) {
    return cin == 3 && cmid == 32 && cout == 64 && conv_stride == 2 && dw_stride == 1 && kh == 3 && kw == 3;
}

int launch_stem_fused(const float* x, const float* cw, const float* cshift, const float* wd, const float* dscale,
                      const float* dshift, const void* wsplit, const float* descale, const float* pshift, float* y, int n,
                      int h, int w, int cpad_t, int cpad_l, int oh, int ow, int a_log2, int act, hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && h >= 3 && w >= 3 && oh > 0 && ow > 0, HSEFR_ERR_INVALID, "stem_fused: bad shape");
    HSEFR_REQUIRE(oh == (h + 1) / 2 && ow == (w + 1) / 2, HSEFR_ERR_INVALID, "stem_fused: %dx%d is not the SAME stride-2 map of %dx%d", oh, ow, h, w);
    HSEFR_REQUIRE(a_log2 > 0 && a_log2 <= 12, HSEFR_ERR_INVALID, "stem_fused: a_log2=%d", a_log2);
    if (n == 0) return HSEFR_OK;
    StemParams p;
    p.x = x; p.cw = cw; p.cshift = cshift; p.wd = (const float4*)wd; p.dscale = (const float4*)dscale;
    p.dshift = (const float4*)dshift; p.wsplit = (const float*)wsplit; p.descale = descale; p.pshift = pshift; p.y = y;
    p.H = h; p.W = w; p.OH = oh; p.OW = ow; p.cpad_t = cpad_t; p.cpad_l = cpad_l;
    p.tiles_w = (ow + TW - 1) / TW; p.tiles_h = (oh + TH - 1) / TH;
    const long long total = (long long)n * p.tiles_w * p.tiles_h;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "stem_fused: grid too large");
    p.total = (unsigned)total;
    p.a_scale = ldexpf(1.f, a_log2);
    p.reverse = sweep_reverse();
    p.stamps = nullptr;
#ifdef HSEFR_STEM_STAMPS
    p.stamps = stamp_buffer(s);
#endif
    const unsigned g = p.total < 512u ? p.total : 512u;      // 512 % 8 == 0: the kernel's incremental patch cursor relies on it
#define HSEFR_STEM(A) HSEFR_LAUNCH((stem_fused_kernel<A>), dim3(g), dim3(256), 0, s, p)
    if (act == HSEFR_ACT_RELU6) HSEFR_STEM(HSEFR_ACT_RELU6);
    else if (act == HSEFR_ACT_RELU) HSEFR_STEM(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_NONE) HSEFR_STEM(HSEFR_ACT_NONE);
    else { set_error("stem_fused: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_STEM
    return launch_status("stem_fused");
}

}  // namespace hsefr
