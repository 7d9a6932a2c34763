// MobileNet's stem AND the depthwise half of its second block in one kernel:
//   conv1 3x3/2 (3 -> 32, + shift + ReLU6) -> depthwise 3x3/1 (+ scale + shift + ReLU6) -> pointwise 1x1 (32 -> 64, + shift +
//   ReLU6) -> depthwise 3x3/2 (+ scale + shift + act),                                              NHWC fp32, gfx950.
//
// Replaces graph nodes #30-#55 (conv1 .. conv_dw_2_relu), run by tf_sess.run at facerec_test.py:120 /
// facial_analysis.py:109.  The 96x96x64 map between the two blocks is the largest tensor of the network (2.36 MB per
// face, 604 MB per 256-face batch): written by one kernel and read by the next it is 18 % of all the bytes the trunk
// moves.  Here the image goes in (0.44 MB per face) and the STRIDED depthwise result comes out (48x48x64, 0.59 MB);
// conv1's map, the first depthwise's and the 96x96x64 one only ever exist patch-wise in LDS.  What follows is the plain
// pointwise GEMM of block 2 (64 -> 128 on 48x48 pixels).
//
// One workgroup (256 threads) per 4 x 8 patch of the output (all 64 channels):
//   region R1 = the 9 x 17 pixels of the 96x96 map that patch's 3x3/2 windows cover; R0 = R1 + the first depthwise's
//   halo = 11 x 19 conv1 pixels (209).
//   A  gather   209 threads fetch the 3x3x3 input window of one conv1 pixel each (9 dwordx3 loads, clamped addresses,
//               padding as 0/1 factors) and write one 128-B im2col row (k = dy*9 + dx*3 + ci, zero-padded to 32);
//   B  conv1    [224 x 32] . [32 x 32] on v_mfma_f32_16x16x4_f32 (exact fp32: the image is unbounded), 28 (row block,
//               channel block) pairs dealt 7 per wave; + shift, ReLU6, pixels outside the map zeroed -> LDS;
//   C  depthwise 1 over R1 from LDS (+ scale + shift + ReLU6), scaled by 2^12 and split into f16 hi + lo -> GEMM A tile;
//   D  pointwise [160 x 32] . [32 x 64] on v_mfma_f32_16x16x32_f16 (al*bh + ah*bl + ah*bh; wave w = channels 16w..16w+15,
//               its weight fragments live in registers); * descale + shift, ReLU6, out-of-map pixels zeroed -> LDS;
//   E  depthwise 2 (stride 2) over that tile from LDS, + scale + shift + act -> 16-B stores (whole 256-B pixels).
// Halo recomputation: 209 conv1 / 153 block-1 pixels per 128 net (1.63x / 1.2x of two cheap layers).  The next patch's
// gather is in flight during B-E.  LDS: im2col rows and the A tile share 28 KB, the conv1 region and the 96x96x64
// patch share 39 KB -> 2 workgroups per CU.
#include <type_traits>

#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };

struct Stem2Params {
    const float* x;        // [N,H,W,3]
    const float* cw;       // conv1 kernel, TF HWIO [3,3,3,32]
    const float* cshift;   // [32]
    const float4* wd1;     // depthwise 1 [9][8] float4
    const float4* d1scale; // [8]
    const float4* d1shift; // [8]
    const float* wsplit;   // pointwise split rows [64][1][64 f16]
    const float* descale;  // [64]
    const float* pshift;   // [64]
    const float4* wd2;     // depthwise 2 [9][16] float4
    const float4* d2scale; // [16]
    const float4* d2shift; // [16]
    float* y;              // [N,OH2,OW2,64]
    int H, W, H1, W1, OH2, OW2, cpad_t, cpad_l, pad_t2, pad_l2, tiles_w, tiles_h;
    unsigned total;
    float a_scale;
    int reverse;
    unsigned long long* stamps;   // diagnostic builds (-DHSEFR_STEM_STAMPS) only
};

constexpr int PH = 4, PW = 8;                         // output patch (of the stride-2 depthwise)
constexpr int R1H = 2 * PH + 1, R1W = 2 * PW + 1;     // block-1 region 9 x 17
constexpr int R1PIX = R1H * R1W;                      // 153
constexpr int R1ROWS = 160;                           // 10 MFMA row blocks of 16
constexpr int R0H = R1H + 2, R0W = R1W + 2;           // conv1 region 11 x 19
constexpr int R0PIX = R0H * R0W;                      // 209
constexpr int R0ROWS = 224;                           // 14 MFMA row blocks of 16
constexpr int COP = 36;                               // floats per pixel row of the conv1 region in LDS (32 + 4): taps sit at
                                                      // compile-time offsets from one base (no per-tap swizzle arithmetic)
constexpr int P1P = 68;                               // floats per pixel row of the 96x96x64 patch in LDS (64 + 4: rows 4 banks apart)

__device__ __forceinline__ int swz32(int row, int chunk) { return row * 32 + 4 * (chunk ^ ((row >> 1) & 7) ^ ((row & 1) << 2)); }       // floats
__device__ __forceinline__ int swzb(int row, int chunk) { return row * 128 + 16 * (chunk ^ ((row >> 1) & 7) ^ ((row & 1) << 2)); }      // bytes; b64 writes of adjacent rows land in different halves of the 128-B bank window
__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
// 4-wide fused multiply-add on vector types: lowers to two v_pk_fma_f32 (same rounding as fmaf, half the instructions);
// used where no MFMA shares the issue slots (the depthwise stages).
__device__ __forceinline__ f32x4 vfma(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x4 as_v(float4 a) { return (f32x4){a.x, a.y, a.z, a.w}; }

template <int ACT>
__global__ __launch_bounds__(256, 2) void stem2_fused_kernel(Stem2Params p) {
    // LDS regions: U1 = im2col rows (stages A-B), then the GEMM A tile (C-D); U2 = conv1 region (B-C), then the
    // 96x96x64 patch (D-E).
    __shared__ __attribute__((aligned(16))) float U1[R0ROWS * 32];          // 28 KB
    __shared__ __attribute__((aligned(16))) float U2[R1PIX * P1P];          // 41 KB (the conv1 region needs 224*32 floats = 28 KB)
    __shared__ __attribute__((aligned(16))) float Cw[32 * 32];              // conv1 weights [n][k]
    __shared__ __attribute__((aligned(16))) float4 W2[9 * 16];              // depthwise-2 weights
    __shared__ float Cv[R0ROWS];                                            // 1 = conv1 pixel inside its map
    __shared__ float Pv[R1ROWS];                                            // 1 = block-1 pixel inside its map
    static_assert(R0ROWS * COP <= R1PIX * P1P, "conv1 region fits in U2");
    float* Ic = U1;
    unsigned char* As = (unsigned char*)U1;
    float* Co = U2;
    float* P1 = U2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, q4 = lane >> 4;

    // ---- constants, once per workgroup ----
    for (int i = tid; i < 32 * 32; i += 256) {
        const int n = i >> 5, k = i & 31;
        Cw[swz32(n, k >> 2) + (k & 3)] = k < 27 ? p.cw[k * 32 + n] : 0.f;
    }
    if (tid < 9 * 16) W2[tid] = p.wd2[tid];
    const int c4l = tid & 7;                         // depthwise-1 channel quad of this thread (256 % 8 == 0: fixed)
    float4 wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = p.wd1[i * 8 + c4l];
    const float4 d1sc = p.d1scale[c4l], d1sh = p.d1shift[c4l];
    const int c4o = tid & 15;                        // depthwise-2 channel quad of this thread (256 % 16 == 0: fixed)
    const float4 d2sc = p.d2scale[c4o], d2sh = p.d2shift[c4o];
    // pointwise: wave w owns channels 16w .. 16w+15; lane (n = 16w + l16, k-slice q4) holds its weight fragments for good
    const f16x8 bh = *(const f16x8*)((const unsigned char*)p.wsplit + (size_t)(wave * 16 + l16) * 128 + 16 * q4);
    const f16x8 bl = *(const f16x8*)((const unsigned char*)p.wsplit + (size_t)(wave * 16 + l16) * 128 + 64 + 16 * q4);
    // epilogue constants of stages B and D: lane owns 4 consecutive channels (accumulator rows 4*q4 .. 4*q4+3)
    f32x4 csh[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) csh[nb] = *(const f32x4*)(p.cshift + nb * 16 + 4 * q4);
    const f32x4 pds = *(const f32x4*)(p.descale + wave * 16 + 4 * q4), psh = *(const f32x4*)(p.pshift + wave * 16 + 4 * q4);

    // ---- patch cursor (advanced with carries: no divisions in the loop) ----
    struct Cur { int n, th, tw; };
    auto decode = [&](unsigned t) {
        const unsigned lt = xcd_remap_dir(t, p.total, p.reverse);
        Cur c;
        c.tw = lt % p.tiles_w;
        c.th = (lt / p.tiles_w) % p.tiles_h;
        c.n = lt / (p.tiles_w * p.tiles_h);
        return c;
    };
    const int stride_lt = gridDim.x / 8;             // launch guarantees gridDim.x % 8 == 0 whenever the kernel loops
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

    // ---- stage A: gather one conv1 pixel's 3x3x3 window, scatter it as an im2col row ----
    F3 g[9];
    int gih0 = 0, giw0 = 0;       // top-left input coordinate of the gathered window (the padding masks are re-derived from it)
    float cvalid = 0.f, pvalid = 0.f;
    auto gather = [&](Cur c) {
        const int y10 = 2 * c.th * PH - p.pad_t2, x10 = 2 * c.tw * PW - p.pad_l2;     // block-1 region origin
        {   // conv1 pixel of this thread (region R0 starts one pixel up/left of R1)
            const int rp = tid < R0PIX ? tid : R0PIX - 1;      // threads 209..255 shadow the last pixel (rows never used)
            const int cy = y10 - 1 + rp / R0W, cx = x10 - 1 + rp % R0W;
            cvalid = (tid < R0PIX && cy >= 0 && cy < p.H1 && cx >= 0 && cx < p.W1) ? 1.f : 0.f;
            const int ih0 = cy * 2 - p.cpad_t, iw0 = cx * 2 - p.cpad_l;
            gih0 = ih0; giw0 = iw0;
            const float* img = p.x + (size_t)c.n * p.H * p.W * 3;       // uniform
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int ihc = min(max(ih0 + dy, 0), p.H - 1);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int iwc = min(max(iw0 + dx, 0), p.W - 1);
                    g[dy * 3 + dx] = *(const F3*)(img + (unsigned)(ihc * p.W + iwc) * 3u);
                }
            }
        }
        {   // validity of block-1 pixel tid of R1 (rows >= 153 are padding rows of the GEMM)
            const int q = tid < R1PIX ? tid : 0;
            const int y1 = y10 + q / R1W, x1 = x10 + q % R1W;
            pvalid = (tid < R1PIX && y1 >= 0 && y1 < p.H1 && x1 >= 0 && x1 < p.W1) ? 1.f : 0.f;
        }
    };
    auto scatter = [&]() {
        float v[32];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int ih = gih0 + q / 3, iw = giw0 + q % 3;
            const float m = (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) ? 1.f : 0.f;
            v[3 * q] = g[q].a * m;
            v[3 * q + 1] = g[q].b * m;
            v[3 * q + 2] = g[q].c * m;
        }
#pragma unroll
        for (int q = 27; q < 32; ++q) v[q] = 0.f;
        if (tid < R0ROWS) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                f32x4 o;
                o[0] = v[4 * c]; o[1] = v[4 * c + 1]; o[2] = v[4 * c + 2]; o[3] = v[4 * c + 3];
                *(f32x4*)(&Ic[swz32(tid, c)]) = o;
            }
            Cv[tid] = cvalid;
        }
        if (tid < R1ROWS) Pv[tid] = pvalid;
    };

    unsigned t = blockIdx.x;
    if (t >= p.total) return;
    Cur cur = decode(t);
    gather(cur);
    scatter();
    __syncthreads();

    STEM_STAMP_DECL;
    while (true) {
        const unsigned tn = t + gridDim.x;
        const bool more = tn < p.total;
        const Cur nxt = advance(cur);
        STEM_STAMP(0);

        // ---- stage B: conv1 on the fp32 MFMA; 14 row blocks x 2 channel blocks = 28 pairs, 7 per wave ----
        auto conv_pairs = [&](auto NP, int first) {     // NP pairs at once: independent accumulators hide the MFMA latency
            constexpr int N = decltype(NP)::value;
            f32x4 acc[N];
#pragma unroll
            for (int i = 0; i < N; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 a[N], b[N];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const int pr = first + i;
                    a[i] = *(const f32x4*)(&Ic[swz32((pr >> 1) * 16 + l16, 4 * s + q4)]);
                    b[i] = *(const f32x4*)(&Cw[swz32((pr & 1) * 16 + l16, 4 * s + q4)]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < N; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i][e], a[i][e], acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                // lane: pixel m = 16*mb + l16, channels nb*16 + 4*q4 + (0..3)   (operands swapped: weights first)
                const int pr = first + i, m = (pr >> 1) * 16 + l16, nb = pr & 1;
                const float valid = Cv[m];
                const f32x4 sh = nb ? csh[1] : csh[0];
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = relu6(acc[i][e] + sh[e]) * valid;
                *(f32x4*)(&Co[m * COP + 4 * (nb * 4 + q4)]) = o;
            }
        };
        conv_pairs(std::integral_constant<int, 3>(), wave * 7);
        conv_pairs(std::integral_constant<int, 2>(), wave * 7 + 3);
        conv_pairs(std::integral_constant<int, 2>(), wave * 7 + 5);
        STEM_STAMP(1);
        __syncthreads();     // conv1 region complete; im2col rows dead
        STEM_STAMP(2);

        // ---- stage C: depthwise 1 over the 153 block-1 pixels, straight from LDS -> split-f16 A tile ----
#pragma unroll 2
        for (int it = 0; it < 5; ++it) {
            const int q = (tid >> 3) + 32 * it;                // block-1 region pixel
            if (q < R1ROWS) {
                const int qq = q < R1PIX ? q : R1PIX - 1;      // rows 153..159: anything finite (their outputs are unused)
                const int ry = qq / R1W, rx = qq % R1W;        // conv1 region pixel (ry + dy, rx + dx)
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* base = &Co[(ry * R0W + rx) * COP + 4 * c4l];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        s = vfma(*(const f32x4*)(base + (dy * R0W + dx) * COP), as_v(wk[dy * 3 + dx]), s);
                const f32x4 o = vfma(s, as_v(d1sc), as_v(d1sh));
                f32x4 v;
                v[0] = relu6(o[0]); v[1] = relu6(o[1]); v[2] = relu6(o[2]); v[3] = relu6(o[3]);
                v = v * p.a_scale;
                const f16x4 hi = __builtin_convertvector(v, f16x4);
                const f16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), f16x4);
                *(f16x4*)(&As[swzb(q, c4l >> 1) + 8 * (c4l & 1)]) = hi;
                *(f16x4*)(&As[swzb(q, 4 + (c4l >> 1)) + 8 * (c4l & 1)]) = lo;
            }
        }
        STEM_STAMP(3);
        __syncthreads();     // A tile complete; conv1 region dead
        STEM_STAMP(2);

        if (more) gather(nxt);       // next patch's window loads fly during stages D-E (their 27 registers are free in B and C)
        STEM_STAMP(7);
        // ---- stage D: pointwise on the f16 MFMA (K = 32 in one instruction); wave w = channels 16w..16w+15, all 10 row blocks
#pragma unroll 2
        for (int mb = 0; mb < 10; ++mb) {
            const f16x8 ah = *(const f16x8*)(&As[swzb(mb * 16 + l16, q4)]);
            const f16x8 al = *(const f16x8*)(&As[swzb(mb * 16 + l16, 4 + q4)]);
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc, 0, 0, 0);
            // lane: block-1 pixel m = 16*mb + l16, channels 16*wave + 4*q4 + (0..3)
            const int m = mb * 16 + l16;
            if (m < R1PIX) {
                const float valid = Pv[m];
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = relu6(fmaf(acc[e], pds[e], psh[e])) * valid;
                *(f32x4*)(&P1[m * P1P + wave * 16 + 4 * q4]) = o;
            }
        }
        STEM_STAMP(4);
        __syncthreads();     // 96x96x64 patch complete; A tile dead
        STEM_STAMP(2);

        // ---- stage E: depthwise 2 (stride 2) from LDS -> global ----
        {
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + (size_t)cur.n * p.OH2 * p.OW2 * 64, (long long)p.OH2 * p.OW2 * 256);
#pragma unroll 1
            for (int it = 0; it < 2; ++it) {
                const int px = (tid >> 4) + 16 * it;           // 0..31: output pixel of the patch
                const int i = px >> 3, j = px & 7;
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        s = vfma(*(const f32x4*)(&P1[((2 * i + dy) * R1W + 2 * j + dx) * P1P + 4 * c4o]), as_v(W2[(dy * 3 + dx) * 16 + c4o]), s);
                const f32x4 o = vfma(s, as_v(d2sc), as_v(d2sh));
                f32x4 v;
                v[0] = apply_act<ACT>(o[0]); v[1] = apply_act<ACT>(o[1]); v[2] = apply_act<ACT>(o[2]); v[3] = apply_act<ACT>(o[3]);
                const int oh = cur.th * PH + i, ow = cur.tw * PW + j;
                // a pixel outside the map gets an offset beyond the resource and the store is dropped (no branch)
                const unsigned voff = (oh < p.OH2 && ow < p.OW2) ? (unsigned)(oh * p.OW2 + ow) * 256u + 16u * c4o : 0x80000000u;
                bstore16(v, ry, voff, 0);
            }
        }
        STEM_STAMP(5);
        STEM_STAMP_COUNT;
        if (!more) break;
        // U1 (im2col rows) is free since the barrier after stage D; U2 is read by stage E of slower waves, but the next
        // writer of U2 is stage B, behind the barrier below
        scatter();
        STEM_STAMP(6);
        __syncthreads();
        STEM_STAMP(2);
        t = tn;
        cur = nxt;
    }
    STEM_STAMP_FLUSH(p.stamps, lane, wave);
}

}  // namespace

bool stem2_fused_supported(int cin, int c1, int c2, int conv_stride, int dw1_stride, int dw2_stride, int kh, int kw) {
    return cin == 3 && c1 == 32 && c2 == 64 && conv_stride == 2 && dw1_stride == 1 && dw2_stride == 2 && kh == 3 && kw == 3;
}

int launch_stem2_fused(const float* x, const float* cw, const float* cshift, const float* wd1, const float* d1scale,
                       const float* d1shift, const void* wsplit, const float* descale, const float* pshift, const float* wd2,
                       const float* d2scale, const float* d2shift, float* y, int n, int h, int w, int cpad_t, int cpad_l,
                       int h1, int w1, int pad_t2, int pad_l2, int oh2, int ow2, int a_log2, int act, hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && h >= 3 && w >= 3 && h1 > 0 && w1 > 0 && oh2 > 0 && ow2 > 0, HSEFR_ERR_INVALID, "stem2_fused: bad shape");
    HSEFR_REQUIRE(h1 == (h + 1) / 2 && w1 == (w + 1) / 2 && oh2 == (h1 + 1) / 2 && ow2 == (w1 + 1) / 2, HSEFR_ERR_INVALID,
                  "stem2_fused: %dx%d -> %dx%d -> %dx%d is not two SAME stride-2 steps", h, w, h1, w1, oh2, ow2);
    HSEFR_REQUIRE(pad_t2 >= 0 && pad_t2 <= 1 && pad_l2 >= 0 && pad_l2 <= 1, HSEFR_ERR_INVALID, "stem2_fused: depthwise-2 padding %d,%d", pad_t2, pad_l2);
    HSEFR_REQUIRE(a_log2 > 0 && a_log2 <= 12, HSEFR_ERR_INVALID, "stem2_fused: a_log2=%d", a_log2);
    if (n == 0) return HSEFR_OK;
    Stem2Params p;
    p.x = x; p.cw = cw; p.cshift = cshift; p.wd1 = (const float4*)wd1; p.d1scale = (const float4*)d1scale;
    p.d1shift = (const float4*)d1shift; p.wsplit = (const float*)wsplit; p.descale = descale; p.pshift = pshift;
    p.wd2 = (const float4*)wd2; p.d2scale = (const float4*)d2scale; p.d2shift = (const float4*)d2shift; p.y = y;
    p.H = h; p.W = w; p.H1 = h1; p.W1 = w1; p.OH2 = oh2; p.OW2 = ow2; p.cpad_t = cpad_t; p.cpad_l = cpad_l;
    p.pad_t2 = pad_t2; p.pad_l2 = pad_l2;
    p.tiles_w = (ow2 + PW - 1) / PW; p.tiles_h = (oh2 + PH - 1) / PH;
    const long long total = (long long)n * p.tiles_w * p.tiles_h;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "stem2_fused: grid too large");
    p.total = (unsigned)total;
    p.a_scale = ldexpf(1.f, a_log2);
    p.reverse = sweep_reverse();
    p.stamps = nullptr;
#ifdef HSEFR_STEM_STAMPS
    p.stamps = stamp_buffer(s);
#endif
    const unsigned g = p.total < 512u ? p.total : 512u;      // 512 % 8 == 0: the kernel's incremental patch cursor relies on it
#define HSEFR_STEM2(A) HSEFR_LAUNCH((stem2_fused_kernel<A>), dim3(g), dim3(256), 0, s, p)
    if (act == HSEFR_ACT_RELU6) HSEFR_STEM2(HSEFR_ACT_RELU6);
    else if (act == HSEFR_ACT_RELU) HSEFR_STEM2(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_NONE) HSEFR_STEM2(HSEFR_ACT_NONE);
    else { set_error("stem2_fused: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_STEM2
    return launch_status("stem2_fused");
}

}  // namespace hsefr
