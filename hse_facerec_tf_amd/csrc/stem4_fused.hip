// MobileNet's stem and the depthwise half of its second block in one kernel -- fourth generation (round 3):
//   conv1 3x3/2 (3 -> 32, + shift + ReLU6) -> depthwise 3x3/1 (+ scale + shift + ReLU6) -> pointwise 1x1 (32 -> 64, + shift +
//   ReLU6) -> depthwise 3x3/2 (+ scale + shift + act),                                              NHWC, gfx950.
//
// Replaces graph nodes #30-#55 (conv1 .. conv_dw_2_relu), run by tf_sess.run at facerec_test.py:120 / facial_analysis.py:109,
// and -- in the uint8 form -- the float conversion, channel reversal and mean subtraction in front of them
// (facerec_test.py:95-106, facial_analysis.py:98-107).  Same patch geometry as stem3_fused.hip (4 x 8 output pixels per
// workgroup step), same stages C-E (depthwise 1, pointwise, depthwise 2: copied, same operation order, same bits for equal
// conv1 results).  What changed is everything in front of them (72.6 M VALU against 3.8 M MFMA wave-instructions,
// the im2col stage alone 3.1-3.5 k of a patch's 15 k cycles):
//
//   * NO im2col.  stem3 cut 209 rows of 27 values out of the input window with scalar LDS reads (27 ds_read_b32 + ~230 VALU
//     per thread: every input value converted 2.25 times) and wrote them back as MFMA rows.  Here the window is converted ONCE,
//     while it is parked: a thread's 16-byte piece of an image row becomes four f16 (hi) + four f16 (lo) and goes to LDS with
//     two ds_write_b64 -- the window in LDS is the image rows themselves, [row][x][rgb] as f16, hi plane and lo plane.
//   * conv1 reads its MFMA operands STRAIGHT FROM THE WINDOW.  Kernel row dy of a conv pixel is 9 consecutive values of one
//     window row (3 pixels x 3 channels), so the K = 27 contraction is laid out as two 32-deep steps whose 8-value lane slices
//     are contiguous, 4-byte-aligned 16-byte LDS reads:
//         step 0, slice dy (0..2): values 0..7 of kernel row dy        step 1, slice dy: value 8 of kernel row dy (+ 7 whose
//         slice 3: weights zero                                         weights are zero: the next pixel's bytes, finite)
//     -- the pattern of stem7x7_pool.hip.  Two steps x three products = 6 MFMAs of 16 cycles per (16 pixels x 16 channels).
//   * Columns outside the image are whole 16-byte pieces (3 * 32 is a multiple of 4): they are masked like rows outside it, by
//     an out-of-range buffer offset that returns zeros -- no per-value masks.
//   * The next patch's window is loaded during stages B-D, converted and parked during stage E (its LDS region is dead from
//     stage B's barrier on): 4 barriers per patch instead of 7.
//   * The bound check (DESIGN.md lesson 24) runs on the f16 hi halves with packed integer maxima: 1 VALU per value.
//
// uint8 form (U8 = true; SURVEY 8f-1 "fused into conv1's input read"): x is the RESIZED image as the decoder's RGB bytes
// [N,H,W,3].  A byte is exact in ONE f16, so the window has one plane and a product is two MFMAs (wh * a, wl * a).  The mean
// and the channel reversal are folded into constants: the weight image is packed channel-reversed, and
//     conv1(u8 - mean) = sum_valid w * u8  -  sum_valid w * mean          (taps on SAME padding contribute neither)
// where the second sum depends only on which taps are valid: 4 cases (pixel in the last conv row / column or not), prepared
// on the host in float64 as 4 shift vectors.  Error: products are exact, the fp32 accumulation runs over values up to 255
// instead of 151 -- the same 2^-24-scale round-off as the fp32 path (tested at the same 2e-6 bar).
//
// Shapes: H % 4 == 0 and W % 4 == 0 (no top / left padding in either stride-2 step, window rows start 8 bytes into a
// 16-byte unit for every patch); anything else takes stem3_fused.hip.
#include <type_traits>

#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) Frag4 { f16x8 v; };      // a 16-byte MFMA fragment at a 4-byte-aligned LDS address

struct Stem4Params {
    const void* x;         // [N,H,W,3] fp32 (preprocessed) or uint8 RGB (U8)
    const void* cw4;       // conv1 split rows in the two-step K layout: [2 steps][32 channels][hi 32 x f16 | lo 32 x f16]
    const float* cdescale; // [32]  2^-(e_n + in_log2)
    const float* cshift;   // [32]; U8: [4][32] = shift - sum_valid w * mean for (last row ? 2 : 0) + (last column ? 1 : 0)
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
    int* overflow;         // set to 1 if an input value breaks the declared bound (may be null; fp32 form only)
    int H, W, H1, W1, OH2, OW2, tiles_w, tiles_h;
    unsigned total;
    float a_scale, in_scale;
    int reverse;
    long long x_bytes;             // size of the whole input tensor
    unsigned long long* stamps;    // diagnostic builds (-DHSEFR_STEM_STAMPS) only
};

constexpr int PH = 4, PW = 8;                         // output patch (of the stride-2 depthwise)
constexpr int R1H = 2 * PH + 1, R1W = 2 * PW + 1;     // block-1 region 9 x 17
constexpr int R1PIX = R1H * R1W;                      // 153
constexpr int R1ROWS = 160;                           // 10 MFMA row blocks of 16
constexpr int R0H = R1H + 2, R0W = R1W + 2;           // conv1 region 11 x 19
constexpr int R0PIX = R0H * R0W;                      // 209
constexpr int R0ROWS = 224;                           // 14 MFMA row blocks of 16
constexpr int RAWH = 2 * R0H + 1;                     // input window: 23 rows x 39 pixels
constexpr int RAWQ = 30;                              // 4-value pieces per window row (2 + 39 * 3 = 119 values -> 120)
constexpr int WRP = 256;                              // bytes per window row and plane in LDS: 120 f16 + 8 that stay zero
constexpr int WPLANE = RAWH * WRP;                    // 5888
constexpr int WSHIFT = 2;                             // a window row starts this many values into its first piece
constexpr int COP = 36;                               // floats per pixel of the conv1 region in LDS (32 + 4)
constexpr int P1P = 68;                               // floats per pixel of the 96x96x64 patch in LDS (64 + 4)

__device__ __forceinline__ int swzb(int row, int chunk) { return row * 128 + 16 * (chunk ^ ((row >> 1) & 7) ^ ((row & 1) << 2)); }
__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
__device__ __forceinline__ f32x4 vfma(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x4 as_v(float4 a) { return (f32x4){a.x, a.y, a.z, a.w}; }

template <int ACT, bool U8>
__global__ __launch_bounds__(256, 2) void stem4_fused_kernel(Stem4Params p) {
    // LDS: U1 = the input window as f16 (hi plane | lo plane: parked in a patch's stage E, read by the next patch's stage B),
    //      then the GEMM A tile (C-D); U2 = conv1 region (B-C), then the 96x96x64 patch (D-E)
    __shared__ __attribute__((aligned(16))) unsigned char U1[R1ROWS * 128];            // 20 KB
    static_assert(2 * WPLANE <= R1ROWS * 128, "the window planes fit in U1");
    unsigned char* Wn = U1;
    __shared__ __attribute__((aligned(16))) float U2[R1PIX * P1P];                      // 41 KB
    __shared__ __attribute__((aligned(16))) float4 W2[9 * 16];                          // depthwise-2 weights
    __shared__ __attribute__((aligned(16))) float4 W1[9 * 8];                           // depthwise-1 weights
    // per-channel constants (read where they are used: as per-thread registers they lived through every stage and spilled)
    __shared__ __attribute__((aligned(16))) float Kc[64 + 64 + 128 + 128];              // conv1 descale | shift, dw1 scale | shift (x 2^a), pointwise descale | shift, dw2 scale | shift
    __shared__ float Cv[R0ROWS];                                                        // 1 = conv1 pixel inside its map
    __shared__ float Pv[R1ROWS];                                                        // 1 = block-1 pixel inside its map
    __shared__ __attribute__((aligned(16))) float Ct[U8 ? 4 * 32 : 4];                  // U8: the four shift vectors
    __shared__ int Cc[U8 ? R0ROWS : 4];                                                 // U8: case of a conv1 pixel, times 32
    static_assert(R0ROWS * COP <= R1PIX * P1P, "conv1 region fits in U2");
    float* Co = U2;
    float* P1 = U2;
    unsigned char* As = U1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, q4 = lane >> 4;

    // ---- constants, once per workgroup ----
    if (tid < 9 * 16) W2[tid] = p.wd2[tid];
    if (tid < 9 * 8) W1[tid] = p.wd1[tid];
    if (U8 && tid < 4 * 32) Ct[tid] = p.cshift[tid];
    // depthwise 1 feeds the split: its scale / shift carry the 2^a_log2 pre-scale (a power of two commutes with every rounding
    // here: relu6(s * sc + sh) * 2^a == clamp(s * (sc 2^a) + sh 2^a, 0, 6 * 2^a) bit for bit)
    if (tid < 32) { Kc[tid] = p.cdescale[tid]; Kc[32 + tid] = p.cshift[tid]; }
    else if (tid < 64) { Kc[32 + tid] = ((const float*)p.d1scale)[tid - 32] * p.a_scale; Kc[64 + tid] = ((const float*)p.d1shift)[tid - 32] * p.a_scale; }
    else if (tid < 128) { Kc[64 + tid] = p.descale[tid - 64]; Kc[128 + tid] = p.pshift[tid - 64]; }
    else if (tid < 192) { Kc[128 + tid] = ((const float*)p.d2scale)[tid - 128]; Kc[192 + tid] = ((const float*)p.d2shift)[tid - 128]; }
    const float cap6 = 6.f * p.a_scale;
    // conv1: lane (n = 16 nb + l16, k-slice q4) holds the weight fragments of both K steps and both channel blocks for good
    f16x8 cwh[2][2], cwl[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            cwh[st][nb] = *(const f16x8*)((const unsigned char*)p.cw4 + (size_t)(st * 32 + nb * 16 + l16) * 128 + 16 * q4);
            cwl[st][nb] = *(const f16x8*)((const unsigned char*)p.cw4 + (size_t)(st * 32 + nb * 16 + l16) * 128 + 64 + 16 * q4);
        }
    }
    // pointwise: wave w owns channels 16w .. 16w+15
    const f16x8 bh = *(const f16x8*)((const unsigned char*)p.wsplit + (size_t)(wave * 16 + l16) * 128 + 16 * q4);
    const f16x8 bl = *(const f16x8*)((const unsigned char*)p.wsplit + (size_t)(wave * 16 + l16) * 128 + 64 + 16 * q4);

    // conv1 operand addresses, patch-invariant: row block ri of this wave, pixel m = 16 (rb0 + ri) + l16 (pixels past the 209th
    // repeat the last one), K slice q4 = kernel row min(q4, 2): byte offset of the slice's first value in the hi plane
    const int rb0 = (wave * 7) >> 1;
    unsigned caddr[4];
#pragma unroll
    for (int ri = 0; ri < 4; ++ri) {
        const int m = min((rb0 + ri) * 16 + l16, R0PIX - 1);
        const int ry = m / R0W, rx = m - ry * R0W;
        caddr[ri] = (unsigned)((2 * ry + min(q4, 2)) * WRP + 2 * WSHIFT + 12 * rx);
    }

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

    // ---- the input window of a patch: 23 rows x 30 four-value pieces (fp32: 16 bytes, uint8: 4 bytes), three per thread ----
    // A window row starts at image column iw0 = 32 tw - 2, i.e. at value 3 (row W + iw0) = 2 (mod 4) of the tensor: piece q of a
    // row holds window values 4 q - 2 .. 4 q + 1 (the first two belong to the pixel on the left and are never read).  Rows
    // outside the image and pieces that lie wholly in columns outside it (3 * 32 = 0 mod 4: a column edge is a piece edge) get
    // an out-of-range offset: the buffer returns zeros -- SAME padding costs no instruction.
    // (row, piece) of a thread's three items are re-derived from an opaque copy of the thread index where they are used: as
    // loop invariants they, and the address arithmetic hipcc hoisted with them, lived through every stage and spilled
    auto item_rq = [&](int k, int& row, int& q) __attribute__((always_inline)) {
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int item = tx + 256 * k;
        row = item / RAWQ;
        q = item - row * RAWQ;
    };
    typedef typename std::conditional<U8, unsigned, f32x4>::type raw_t;
    raw_t rawv[3];
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
    constexpr int VB = U8 ? 1 : 4;                    // bytes per value
    auto load_window = [&](const Cur& c) {
        const int ih0 = 4 * c.th * PH - 2, iw0 = 4 * c.tw * PW - 2;
        const int q_hi = (3 * (p.W - iw0) + WSHIFT) >> 2;        // first piece wholly right of the image
        const int q_lo = iw0 < 0 ? (3 * (-iw0) + WSHIFT) >> 2 : 0;  // pieces wholly left of it
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int row, q;
            item_rq(k, row, q);
            const int ih = ih0 + row;
            const bool ok = row < RAWH && ih >= 0 && ih < p.H && q >= q_lo && q < q_hi;
            const int first = ((c.n * p.H + ih) * p.W + iw0) * 3 - WSHIFT + 4 * q;   // value index, = 0 (mod 4); < 2^29 (launcher)
            const unsigned voff = ok ? (unsigned)(first * VB) : 0x80000000u;
            if constexpr (U8) rawv[k] = __builtin_amdgcn_raw_buffer_load_b32(rx, voff, 0, 0);
            else rawv[k] = bload16(rx, voff, 0);
        }
    };
    u16x2 amax_pk = {0, 0};
    auto park_window = [&]() {
        // the last 16 bytes of every window row are written by no piece (and the A tile passes through this memory): zero, so
        // that the step-1 reads of a row's last pixels meet finite bytes under their zero weights
        if (threadIdx.x < RAWH * (U8 ? 1 : 2)) {
            float z = 0.f;
            asm volatile("" : "+v"(z));       // (a hoisted zero vector is what hipcc spilled here)
            *(f32x4*)(&Wn[threadIdx.x * WRP + WRP - 16]) = (f32x4){z, z, z, z};
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int row, q;
            item_rq(k, row, q);
            if (row < RAWH) {
                unsigned char* dst = &Wn[row * WRP + 8 * q];
                if constexpr (U8) {
                    const unsigned b = rawv[k];
                    f32x4 v;
                    v[0] = (float)(b & 255u); v[1] = (float)((b >> 8) & 255u); v[2] = (float)((b >> 16) & 255u); v[3] = (float)(b >> 24);
                    *(f16x4*)dst = __builtin_convertvector(v, f16x4);               // exact
                } else {
                    const f32x4 v = rawv[k] * p.in_scale;                            // a power of two: exact
                    const f16x4 hi = __builtin_convertvector(v, f16x4);
                    const f16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), f16x4);
                    *(f16x4*)dst = hi;
                    *(f16x4*)(dst + WPLANE) = lo;
                    // bound check on the hi halves: |v| < 2^15 <=> (bits & 0x7fff) < 0x7800; an Inf / NaN has all exponent bits set
                    const u32x2 hb = __builtin_bit_cast(u32x2, hi);
                    amax_pk = __builtin_elementwise_max(amax_pk, __builtin_bit_cast(u16x2, hb.x & 0x7FFF7FFFu));
                    amax_pk = __builtin_elementwise_max(amax_pk, __builtin_bit_cast(u16x2, hb.y & 0x7FFF7FFFu));
                }
            }
        }
    };
    // validity of the conv1 / block-1 pixels of a patch (read only by patches that touch the map's border)
    auto border_tables = [&](const Cur& c, int t) {
        const int y10 = 2 * c.th * PH, x10 = 2 * c.tw * PW;
        if (t < R0ROWS) {
            const int rp = t < R0PIX ? t : R0PIX - 1;
            const int ry = rp / R0W, rxx = rp - ry * R0W;
            const int cy = y10 - 1 + ry, cx = x10 - 1 + rxx;
            Cv[t] = (t < R0PIX && cy >= 0 && cy < p.H1 && cx >= 0 && cx < p.W1) ? 1.f : 0.f;
            if constexpr (U8) Cc[t] = 32 * ((cy == p.H1 - 1 ? 2 : 0) + (cx == p.W1 - 1 ? 1 : 0));
        }
        if (t < R1ROWS) {
            const int q = t < R1PIX ? t : 0;
            const int y1 = y10 + q / R1W, x1 = x10 + q % R1W;
            Pv[t] = (t < R1PIX && y1 >= 0 && y1 < p.H1 && x1 >= 0 && x1 < p.W1) ? 1.f : 0.f;
        }
    };

    unsigned t = blockIdx.x;
    if (t >= p.total) return;
    Cur cur = decode(t);
    load_window(cur);
    park_window();
    border_tables(cur, tid);
    __syncthreads();

    STEM_STAMP_DECL;
    while (true) {
        const unsigned tn = t + gridDim.x;
        const bool more = tn < p.total;
        const Cur nxt = advance(cur);
        const int y10 = 2 * cur.th * PH, x10 = 2 * cur.tw * PW;
        // a patch whose conv1 region lies inside the map has no pixel to zero: the validity factors are all 1 (uniform test);
        // U8: ... and strictly inside its last row / column, where the taps on the padding change the folded mean term
        const bool interior = y10 - 1 >= 0 && x10 - 1 >= 0 && y10 - 1 + R0H <= p.H1 - (U8 ? 1 : 0) && x10 - 1 + R0W <= p.W1 - (U8 ? 1 : 0);
        STEM_STAMP(0);
        // The thread index is made opaque once per patch: every stage's LDS addresses are then re-derived (a few VALU) instead
        // of being hoisted out of the loop as ~100 loop-invariant VGPRs -- which had the compiler spill to scratch.
        int tix = threadIdx.x;
        asm volatile("" : "+v"(tix));
        const int tid = tix, lane = tid & 63, l16 = lane & 15, q4 = lane >> 4, c4l = tid & 7, c4o = tid & 15;
        STEM_STAMP(7);

        // ---- stage B: conv1 straight from the window; 14 row blocks x 2 channel blocks = 28 pairs, 7 per wave ----
        // pair pr = 7 * wave + i covers row block pr >> 1, channel block pr & 1: the (row block, channel block) of slot i
        // depends on the parity of the wave only -- two unrolled variants, register indices all static
        auto conv_stage = [&](auto ODDC) __attribute__((always_inline)) {
            constexpr int ODD = decltype(ODDC)::value;
            // all fragments of the wave's four row blocks first (one LDS round trip), then the MFMAs with the seven accumulators
            // interleaved: consecutive MFMAs never depend on each other
            f16x8 ah[2][4], al[2][4];
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const unsigned char* a0 = Wn + caddr[ri];
                ah[0][ri] = ((const Frag4*)(a0))->v;
                ah[1][ri] = ((const Frag4*)(a0 + 16))->v;
                if constexpr (!U8) { al[0][ri] = ((const Frag4*)(a0 + WPLANE))->v; al[1][ri] = ((const Frag4*)(a0 + WPLANE + 16))->v; }
            }
            f32x4 acc[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int pdt = U8 ? 1 : 0; pdt < 3; ++pdt)
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        constexpr int dummy = 0; (void)dummy;
                        const int ri = (i + ODD) >> 1, nb = (i + ODD) & 1;
                        // per K step the products in the order (wh*al, wl*ah, wh*ah) of the pointwise kernels; a byte has no lo term
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pdt == 1 ? cwl[st][nb] : cwh[st][nb], pdt == 0 ? al[st][ri] : ah[st][ri], acc[i], 0, 0, 0);
                    }
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                // lane: pixel m = 16 * rb + l16, channels nb*16 + 4*q4 + (0..3)   (operands swapped: weights first)
                const int ri = (i + ODD) >> 1, nb = (i + ODD) & 1;
                const int m = (rb0 + ri) * 16 + l16;
                const f32x4 ds = *(const f32x4*)(&Kc[nb * 16 + 4 * q4]);
                f32x4 sh = *(const f32x4*)(&Kc[32 + nb * 16 + 4 * q4]);
                if constexpr (U8) { if (!interior) sh = *(const f32x4*)(&Ct[Cc[m] + nb * 16 + 4 * q4]); }
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = relu6(fmaf(acc[i][e], ds[e], sh[e]));
                if (!interior) o = o * Cv[m];
                *(f32x4*)(&Co[m * COP + 4 * (nb * 4 + q4)]) = o;
            }
        };
        if (wave & 1) conv_stage(std::integral_constant<int, 1>());
        else conv_stage(std::integral_constant<int, 0>());
        STEM_STAMP(3);
        __syncthreads();     // conv1 region complete; the window is dead
        STEM_STAMP(2);
        if (more) load_window(nxt);      // next patch's window: in flight during stages C-D, parked in stage E

        // ---- stage C: depthwise 1.  Thread = (channel quad, run of <= 6 pixels of one region row): 3 x 8 taps read once ----
        {
            const int grp = tid >> 3;                              // 27 runs: row = grp / 3, columns 6 * (grp % 3) ..
            if (grp < 27) {
                const int ry = grp / 3, c0 = 6 * (grp - 3 * ry);
                // row by row: 8 taps of a region row feed 6 running sums (the products of a pixel are added in the order
                // dy = 0 (dx 0,1,2), dy = 1, dy = 2 of stem2_fused.hip / dwconv.hip: same bits)
                f32x4 sum[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) sum[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    f32x4 tap[8];
#pragma unroll
                    // (the last run is 5 pixels wide: its eighth tap is the next row's first pixel, read and never used --
                    // all 24 addresses are one base plus a constant)
                    for (int col = 0; col < 8; ++col) tap[col] = *(const f32x4*)(&Co[((ry + dy) * R0W + c0 + col) * COP + 4 * c4l]);
                    const f32x4 w0 = as_v(W1[(dy * 3 + 0) * 8 + c4l]), w1 = as_v(W1[(dy * 3 + 1) * 8 + c4l]), w2 = as_v(W1[(dy * 3 + 2) * 8 + c4l]);
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        sum[j] = vfma(tap[j], w0, sum[j]);
                        sum[j] = vfma(tap[j + 1], w1, sum[j]);
                        sum[j] = vfma(tap[j + 2], w2, sum[j]);
                    }
                }
                const f32x4 d1sc = *(const f32x4*)(&Kc[64 + 4 * c4l]), d1sh = *(const f32x4*)(&Kc[96 + 4 * c4l]);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    if (c0 + j < R1W) {
                        const f32x4 o = vfma(sum[j], d1sc, d1sh);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(o[e], 0.f), cap6);
                        const f16x4 hi = __builtin_convertvector(v, f16x4);
                        const f16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), f16x4);
                        const int q = ry * R1W + c0 + j;
                        *(f16x4*)(&As[swzb(q, c4l >> 1) + 8 * (c4l & 1)]) = hi;
                        *(f16x4*)(&As[swzb(q, 4 + (c4l >> 1)) + 8 * (c4l & 1)]) = lo;
                    }
                }
            }
        }
        STEM_STAMP(4);
        __syncthreads();     // A tile complete (rows 153..159 hold stale bytes: their products are never stored); conv1 region dead
        STEM_STAMP(2);

        // ---- stage D: pointwise on the f16 MFMA (K = 32 in one instruction); wave w = channels 16w..16w+15, all 10 row blocks
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f16x8 ah[5], al[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                ah[i] = *(const f16x8*)(&As[swzb((5 * half + i) * 16 + l16, q4)]);
                al[i] = *(const f16x8*)(&As[swzb((5 * half + i) * 16 + l16, 4 + q4)]);
            }
            f32x4 acc[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pdt = 0; pdt < 3; ++pdt)
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pdt == 1 ? bl : bh, pdt == 0 ? al[i] : ah[i], acc[i], 0, 0, 0);
            const f32x4 pds = *(const f32x4*)(&Kc[128 + wave * 16 + 4 * q4]), psh = *(const f32x4*)(&Kc[192 + wave * 16 + 4 * q4]);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                // lane: block-1 pixel m = 16*mb + l16, channels 16*wave + 4*q4 + (0..3)
                const int m = (5 * half + i) * 16 + l16;
                if (m < R1PIX) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = relu6(fmaf(acc[i][e], pds[e], psh[e]));
                    if (!interior) o = o * Pv[m];
                    *(f32x4*)(&P1[m * P1P + wave * 16 + 4 * q4]) = o;
                }
            }
        }
        STEM_STAMP(5);
        __syncthreads();     // 96x96x64 patch complete; A tile dead
        STEM_STAMP(2);

        // the next patch's window (loaded during B-D) and border tables first: their LDS is dead since stage B / D of this patch,
        // and the piece registers are free before the depthwise's taps need them
        if (more) {
            park_window();
            const int ny = 2 * nxt.th * PH, nx = 2 * nxt.tw * PW;
            if (!(ny - 1 >= 0 && nx - 1 >= 0 && ny - 1 + R0H <= p.H1 - (U8 ? 1 : 0) && nx - 1 + R0W <= p.W1 - (U8 ? 1 : 0))) border_tables(nxt, tid);
        }
        STEM_STAMP(1);
        // ---- stage E: depthwise 2 (stride 2) from LDS -> global; both output pixels of a thread in flight together ----
        {
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + (size_t)cur.n * p.OH2 * p.OW2 * 64, (long long)p.OH2 * p.OW2 * 256);
            const f32x4 d2sc = *(const f32x4*)(&Kc[256 + 4 * c4o]), d2sh = *(const f32x4*)(&Kc[320 + 4 * c4o]);
            f32x4 tp[2][9];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int px = (tid >> 4) + 16 * it, i = px >> 3, j = px & 7;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) tp[it][dy * 3 + dx] = *(const f32x4*)(&P1[((2 * i + dy) * R1W + 2 * j + dx) * P1P + 4 * c4o]);
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int px = (tid >> 4) + 16 * it, i = px >> 3, j = px & 7;
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 9; ++k) s = vfma(tp[it][k], as_v(W2[k * 16 + c4o]), s);
                const f32x4 o = vfma(s, d2sc, d2sh);
                f32x4 v;
                v[0] = apply_act<ACT>(o[0]); v[1] = apply_act<ACT>(o[1]); v[2] = apply_act<ACT>(o[2]); v[3] = apply_act<ACT>(o[3]);
                const int oh = cur.th * PH + i, ow = cur.tw * PW + j;
                // a pixel outside the map gets an offset beyond the resource and the store is dropped (no branch)
                const unsigned voff = (oh < p.OH2 && ow < p.OW2) ? (unsigned)(oh * p.OW2 + ow) * 256u + 16u * c4o : 0x80000000u;
                bstore16(v, ry, voff, 0);
            }
        }
        STEM_STAMP(6);
        STEM_STAMP_COUNT;
        if (!more) break;
        __syncthreads();     // the 96x96x64 patch is dead; window and tables of the next patch complete
        t = tn;
        cur = nxt;
    }
    if constexpr (!U8) {
        const unsigned am = max((unsigned)amax_pk[0], (unsigned)amax_pk[1]);
        if (am >= 0x7800u && p.overflow) atomicOr(p.overflow, 1);
    }
    STEM_STAMP_FLUSH(p.stamps, (int)(threadIdx.x & 63), wave);
}

}  // namespace

HSEFR_KNOB(g_stem4_grid, 512);      // dev builds: persistent workgroups of the launch (256 = one per CU: latency experiments)
#ifdef HSEFR_DEV
void set_stem4_grid(int v) { g_stem4_grid = v; }
#endif

bool stem4_fused_supported(int cin, int c1, int c2, int conv_stride, int dw1_stride, int dw2_stride, int kh, int kw, int h, int w) {
    return cin == 3 && c1 == 32 && c2 == 64 && conv_stride == 2 && dw1_stride == 1 && dw2_stride == 2 && kh == 3 && kw == 3 &&
           h >= 4 && w >= 4 && h % 4 == 0 && w % 4 == 0;
}

int launch_stem4_fused(const void* x, int x_is_u8, const void* cw4, const float* cdescale, const float* cshift, const float* wd1,
                       const float* d1scale, const float* d1shift, const void* wsplit, const float* descale, const float* pshift,
                       const float* wd2, const float* d2scale, const float* d2shift, float* y, int* overflow, int n, int h, int w,
                       int in_log2, int a_log2, int act, hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && h >= 4 && w >= 4 && h % 4 == 0 && w % 4 == 0, HSEFR_ERR_INVALID,
                  "stem4_fused: %dx%d input (both edges must be multiples of 4; other sizes take stem3_fused)", h, w);
    HSEFR_REQUIRE(a_log2 > 0 && a_log2 <= 12, HSEFR_ERR_INVALID, "stem4_fused: a_log2=%d", a_log2);
    HSEFR_REQUIRE(x_is_u8 ? in_log2 == 0 : (in_log2 >= -8 && in_log2 <= 14), HSEFR_ERR_INVALID, "stem4_fused: in_log2=%d", in_log2);
    HSEFR_REQUIRE((long long)n * h * w * 12 < (1ll << 31) - 64, HSEFR_ERR_UNSUPPORTED,
                  "stem4_fused: the input batch must stay below 2 GB (its offsets travel in 32 bits, 2^31 marks a masked piece)");
    if (n == 0) return HSEFR_OK;
    Stem4Params p;
    p.x = x; p.cw4 = cw4; p.cdescale = cdescale; p.cshift = cshift; p.wd1 = (const float4*)wd1; p.d1scale = (const float4*)d1scale;
    p.d1shift = (const float4*)d1shift; p.wsplit = (const float*)wsplit; p.descale = descale; p.pshift = pshift;
    p.wd2 = (const float4*)wd2; p.d2scale = (const float4*)d2scale; p.d2shift = (const float4*)d2shift; p.y = y; p.overflow = overflow;
    p.H = h; p.W = w; p.H1 = h / 2; p.W1 = w / 2; p.OH2 = h / 4; p.OW2 = w / 4;
    p.tiles_w = (p.OW2 + PW - 1) / PW; p.tiles_h = (p.OH2 + PH - 1) / PH;
    const long long total = (long long)n * p.tiles_w * p.tiles_h;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "stem4_fused: grid too large");
    p.total = (unsigned)total;
    p.a_scale = ldexpf(1.f, a_log2);
    p.in_scale = ldexpf(1.f, in_log2);
    p.reverse = sweep_reverse();
    p.x_bytes = (long long)n * h * w * 3 * (x_is_u8 ? 1 : 4);
    p.stamps = nullptr;
#ifdef HSEFR_STEM_STAMPS
    p.stamps = stamp_buffer(s);
#endif
    const unsigned g = p.total < (unsigned)g_stem4_grid ? p.total : (unsigned)g_stem4_grid;      // % 8 == 0: the kernel's incremental patch cursor relies on it
#define HSEFR_STEM4(A)                                                                                   \
    do {                                                                                                 \
        if (x_is_u8) HSEFR_LAUNCH((stem4_fused_kernel<A, true>), dim3(g), dim3(256), 0, s, p);     \
        else HSEFR_LAUNCH((stem4_fused_kernel<A, false>), dim3(g), dim3(256), 0, s, p);            \
    } while (0)
    if (act == HSEFR_ACT_RELU6) HSEFR_STEM4(HSEFR_ACT_RELU6);
    else if (act == HSEFR_ACT_RELU) HSEFR_STEM4(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_NONE) HSEFR_STEM4(HSEFR_ACT_NONE);
    else { set_error("stem4_fused: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_STEM4
    return launch_status("stem4_fused");
}

}  // namespace hsefr
