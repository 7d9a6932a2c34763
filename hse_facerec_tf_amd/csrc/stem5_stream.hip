// MobileNet's stem and the depthwise half of its second block -- fifth generation (round 4): a STREAMING kernel.
//   conv1 3x3/2 (3 -> 32, + shift + ReLU6) -> depthwise 3x3/1 (+ scale + shift + ReLU6) -> pointwise 1x1 (32 -> 64, + shift +
//   ReLU6) -> depthwise 3x3/2 (+ scale + shift + act),                                              NHWC, gfx950.
//
// Replaces graph nodes #30-#55 (conv1 .. conv_dw_2_relu), run by tf_sess.run at facerec_test.py:120 / facial_analysis.py:109,
// and -- in the uint8 form -- the float conversion, channel reversal and mean subtraction in front of them
// (facerec_test.py:95-106, facial_analysis.py:98-107).  Same arithmetic as stem4_fused.hip: the same MFMA operand layouts,
// the same product order, every depthwise output the same dy = 0 (dx 0, 1, 2), dy = 1, dy = 2 chain -- the same bits.
//
// What changed is the decomposition (PMC of stem4: vector ALU 42 % busy + LDS 48 % + matrix pipe 13 % = 100 %:
// four waves in lock step through four barrier-separated stages use ONE unit at a time, 1.6x halo recompute on top):
//
//   * ONE WAVE = ONE STRIP.  A wave owns KS = 6 output columns of one image and walks DOWN the image, one output row per step.
//     It shares nothing with the other waves of its workgroup but the read-only constants: no barrier after the prologue,
//     no lock step -- the eight waves of a CU (S5_WGS = 2 workgroups x 4) drift apart and the matrix pipe, the vector ALU and
//     the LDS work for different waves at the same time.
//   * NO VERTICAL HALO.  A step brings in 4 new input rows (+ the one it shares with the previous step), computes the TWO new
//     conv1 rows, and every depthwise output is accumulated IN REGISTERS as its three input rows arrive: a conv1 row is read
//     from LDS once (3 taps per output column) and added into the three depthwise rows it belongs to; a pointwise row likewise
//     feeds the stride-2 depthwise row above and below it.  LDS holds line buffers of two rows, not patches: 16.8 KB per wave.
//     conv1 is computed on CXW / 2 KS = 15 / 12 = 1.25x the pixels (horizontal halo only; the 4 x 8 patches of stem4: 1.63x),
//     depthwise 1 and the pointwise on PXW / 2 KS = 13 / 12 (1.08x).
//   * Two start-up steps per strip (their output row is dropped) fill the accumulators: 50 steps for 48 output rows.
//
// Shapes: H % 4 == 0 and W % 4 == 0 (SAME padding then pads bottom / right only in both stride-2 layers), as stem4.
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

struct Stem5Params {
    const void* x;         // [N,H,W,3] fp32 (preprocessed) or uint8 RGB (U8)
    const void* cw4;       // conv1 split rows in the two-step K layout of stem4_fused.hip
    const float* cdescale; // [32]
    const float* cshift;   // [32]; U8: [4][32]
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
    int* overflow;
    int H, W, H1, W1, OH2, OW2;
    int strips, segs, seg_rows;    // strips per image row, vertical segments per strip, output rows per segment
    unsigned total;                // units = N * strips * segs
    float a_scale, in_scale;
    int reverse;
    long long x_bytes;
};

#ifndef S5_LDAUX
#define S5_LDAUX 0     // cache policy of the image loads (buffer aux bits: 1 glc, 2 slc); slc measured in the network: +6 us at 192 x 192 x 256, +-0 at 224 x 224 x 512 (ResNet stem: a gain, stem7s_stream.hip)
#endif
#ifndef S5_KS
#define S5_KS 6
#endif
constexpr int KS = S5_KS;                 // output columns of a strip (48 = 8 x 6: one strip per wave at batch 256 x 192)
constexpr int PXW = 2 * KS + 1;           // block-1 (depthwise 1 / pointwise) columns: 13
constexpr int CXW = 2 * KS + 3;           // conv1 columns: 15
constexpr int WROWS = 5;                  // input rows of a step
constexpr int WPIECES = (2 + (4 * KS + 7) * 3 + 3) / 4;   // 4-value pieces per window row: 2 + 31 * 3 = 95 values -> 24
constexpr int WDQ = WPIECES / 2;          // a lane converts two adjacent pieces
static_assert(WPIECES % 2 == 0 && WROWS * WDQ <= 64, "one lane per pair of pieces");
constexpr int WRP = WPIECES * 8 + 16;     // bytes per window row and plane: the pieces as f16 + 16 that stay zero
constexpr int WPLANE = WROWS * WRP;
constexpr int WSHIFT = 2;                 // a window row starts this many values into its first piece
// Round 5: the conv1 and pointwise rows in LDS are DENSE (128 / 256 bytes per pixel) with their 16-byte chunks XOR-swizzled by
// pkey(pixel) = ((pixel & 1) << 2) | ((pixel >> 1) & 3).  The padded pitches of round 4 (36 / 68 floats) kept the MFMA epilogues'
// 16-byte stores conflict-free but made every depthwise tap read -- 4 pixels x 4 channel quads per ds_read_b128 lane group -- a
// 7-pass access where 4 is the floor (PMC: 18.6 % of the kernel's LDS cycles were bank conflicts, and the LDS is this kernel's
// binding unit; no additive pitch serves both sides: tools' bank model, DESIGN.md lesson 48).  With the swizzle both sides are
// conflict-free for EVERY tap: two pixels that share a lane group's half of the bank window are 2 apart, and pkey keeps their
// bit 2 equal.  The key of pixel p + tap depends on (p + tap) & 7 only, so a lane keeps FOUR (depthwise 1) / SIX (depthwise 2)
// pre-swizzled chunk offsets for good and every tap / round / row displacement is an instruction immediate.
constexpr int COP = 32;                   // floats per conv1 pixel in LDS
constexpr int P1P = 64;                   // floats per pointwise pixel in LDS
__device__ __forceinline__ int pkey(int px) { return ((px & 1) << 2) | ((px >> 1) & 3); }
constexpr int CMB = (2 * CXW + 15) / 16;  // MFMA row blocks of the conv1 GEMM (2 x 15 pixels -> 2)
constexpr int PMB = (2 * PXW + 15) / 16;  // ... of the pointwise GEMM (2 x 13 -> 2)
constexpr int NR1 = (PXW + 7) / 8;        // depthwise-1 rounds: 8 columns x 8 channel quads per round
constexpr int NR2 = (KS + 3) / 4;         // depthwise-2 rounds: 4 output columns x 16 channel quads per round
constexpr int OFF_CO = (2 * WPLANE + 127) / 128 * 128;      // (128-byte aligned regions: a channel block / key bit is then an XOR of the address)
constexpr int OFF_AS = OFF_CO + 2 * CXW * COP * 4;
constexpr int OFF_P1 = OFF_AS + 2 * PXW * 128;         // the A tile's last rows (read, never used) alias the first of these
constexpr int WAVE_LDS = OFF_P1 + 2 * PXW * P1P * 4;
constexpr int WAVES = 4;
#ifndef S5_WGS
#define S5_WGS 2
#endif
static_assert(OFF_AS + PMB * 16 * 128 <= WAVE_LDS, "the A tile's over-read stays inside the wave's own region");
static_assert(4 + 12 * (CXW - 1) + 32 <= WRP, "the second K step of a row's last pixel stays inside the row");
static_assert(S5_WGS * (WAVES * WAVE_LDS + 14 * 1024) <= 160 * 1024, "S5_WGS workgroups per CU");

__device__ __forceinline__ int swzb(int row, int chunk) { return row * 128 + 16 * (chunk ^ ((row >> 1) & 7) ^ ((row & 1) << 2)); }
[[maybe_unused]] __device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
__device__ __forceinline__ f32x4 vfma(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x4 as_v(float4 a) { return (f32x4){a.x, a.y, a.z, a.w}; }
// the waves of a workgroup do not synchronise: what one lane wrote to LDS another lane of the SAME wave reads back, in
// program order (the LDS serves a wave's instructions in order) -- the compiler only has to keep that order
__device__ __forceinline__ void wave_order() { asm volatile("" ::: "memory"); }

template <int ACT, bool U8>
__global__ __launch_bounds__(64 * WAVES, S5_WGS) void stem5_stream_kernel(Stem5Params p) {
    __shared__ __attribute__((aligned(128))) unsigned char Lw[WAVES * WAVE_LDS];
    static_assert(OFF_CO % 128 == 0 && OFF_AS % 128 == 0 && OFF_P1 % 128 == 0 && WAVE_LDS % 128 == 0, "swizzled regions are 128-byte aligned");
    __shared__ __attribute__((aligned(16))) unsigned char Wp[64 * 128];   // the pointwise split rows (8 KB): read per step, see below
    __shared__ __attribute__((aligned(16))) float4 W2[9 * 16];
    __shared__ __attribute__((aligned(16))) float4 W1[9 * 8];
    __shared__ __attribute__((aligned(16))) float Kc[64 + 64 + 128 + 128];   // conv1 descale | shift, dw1 scale | shift (x 2^a), pointwise descale | shift, dw2 scale | shift
    __shared__ __attribute__((aligned(16))) float Ct[U8 ? 4 * 32 : 4];       // U8: the four mean-folded shift vectors

    // (the wave index is pinned to a scalar: everything per strip -- image, columns, step counter, row tests -- then lives in
    // SGPRs and the step loop's control flow is uniform; as `threadIdx.x >> 6` hipcc kept all of it per lane, behind EXEC masks)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, q4 = lane >> 4;

    // ---- constants, once per workgroup (the only barrier of the kernel) ----
    // (16-byte chunk c of row r sits at chunk c ^ (r & 7): a fragment read -- 16 rows x one chunk column at a 128-byte pitch --
    // would otherwise put four lanes of every read group on the same four banks)
#pragma unroll
    for (int i = tid; i < 512; i += 256) ((f32x4*)Wp)[(i & ~7) | ((i & 7) ^ ((i >> 3) & 7))] = ((const f32x4*)p.wsplit)[i];
    if (tid < 9 * 16) W2[tid] = p.wd2[tid];
    if (tid < 9 * 8) W1[tid] = p.wd1[tid];
    if (U8 && tid < 4 * 32) Ct[tid] = p.cshift[tid];
    if (tid < 32) { Kc[tid] = p.cdescale[tid]; Kc[32 + tid] = p.cshift[tid]; }
    else if (tid < 64) { Kc[32 + tid] = ((const float*)p.d1scale)[tid - 32] * p.a_scale; Kc[64 + tid] = ((const float*)p.d1shift)[tid - 32] * p.a_scale; }
    else if (tid < 128) { Kc[64 + tid] = p.descale[tid - 64]; Kc[128 + tid] = p.pshift[tid - 64]; }
    else if (tid < 192) { Kc[128 + tid] = ((const float*)p.d2scale)[tid - 128]; Kc[192 + tid] = ((const float*)p.d2shift)[tid - 128]; }
    unsigned char* const L = Lw + wave * WAVE_LDS;
    // the last 16 bytes of every window row (both planes) are written by no piece: zero, so that the second K step of a row's
    // last pixels meets finite bytes under its zero weights
    if (lane < 2 * WROWS) *(f32x4*)(L + lane * WRP + WRP - 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_assert(WPIECES * 8 == WRP - 16, "pieces fill a row up to its zero tail");
    __syncthreads();
    const float cap6 = 6.f * p.a_scale;

    // conv1 weights: lane (n = 16 nb + l16, k-slice q4), both K steps, both channel blocks, in registers for good.  The pointwise
    // weights (all four channel blocks: a wave computes all 64 channels of its pixels) are READ FROM LDS in every step: held in
    // registers too (32 more) the kernel spilled a dozen lane constants, and their reloads -- vector-memory operations queued
    // behind the step's row requests -- made every step wait for HBM
    f16x8 cwh[2][2], cwl[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            cwh[st][nb] = *(const f16x8*)((const unsigned char*)p.cw4 + (size_t)(st * 32 + nb * 16 + l16) * 128 + 16 * q4);
            cwl[st][nb] = *(const f16x8*)((const unsigned char*)p.cw4 + (size_t)(st * 32 + nb * 16 + l16) * 128 + 64 + 16 * q4);
        }

    // ---- lane roles (the same for every strip) ----
    // conv1 GEMM, row block mb: pixel m = 16 mb + l16 of the 2 x 11 new conv pixels (pixels past the 22nd repeat the last one:
    // same operands, same result, same LDS address); K slice q4 = kernel row min(q4, 2)
    int c_ry[CMB], c_rx[CMB];
    unsigned caddr[CMB], coaddr[CMB];
#pragma unroll
    for (int mb = 0; mb < CMB; ++mb) {
        const int m = min(16 * mb + l16, 2 * CXW - 1);
        c_ry[mb] = m >= CXW ? 1 : 0;
        c_rx[mb] = m - CXW * c_ry[mb];
        caddr[mb] = (unsigned)((2 * c_ry[mb] + min(q4, 2)) * WRP + 2 * WSHIFT + 12 * c_rx[mb]);
        // chunk (4 nb + q4) ^ pkey(pixel): channel block nb lands at this ^ (64 nb)
        coaddr[mb] = (unsigned)(OFF_CO + m * COP * 4 + 16 * (q4 ^ (pkey(m) & 3)) + 64 * (pkey(m) >> 2));
    }
    // pointwise GEMM, row block mb: pixel m = 16 mb + l16 of the 2 x PXW new block-1 pixels
    int p_pr[PMB], p_x[PMB];
#pragma unroll
    for (int mb = 0; mb < PMB; ++mb) {
        const int m = min(16 * mb + l16, 2 * PXW - 1);
        p_pr[mb] = m >= PXW ? 1 : 0;
        p_x[mb] = m - PXW * p_pr[mb];
    }
    const bool p_store_last = 16 * (PMB - 1) + l16 < 2 * PXW;      // the last row block is partly empty
    unsigned paddr[PMB];                                           // pointwise pixel's row, chunk (4 cb + q4) ^ pkey(pixel): + 64 cb, ^ 64 (pkey >> 2)
#pragma unroll
    for (int mb = 0; mb < PMB; ++mb) {
        const int m = p_pr[mb] * PXW + p_x[mb];
        paddr[mb] = (unsigned)(OFF_P1 + m * P1P * 4 + 16 * (q4 ^ (pkey(m) & 3)) + 64 * (pkey(m) >> 2));
    }
    // window pieces: lane = WDQ row + dq loads pieces 2 dq and 2 dq + 1 of input row `row`
    const int wl_row = lane / WDQ, wl_dq = lane - WDQ * wl_row;
    const bool wl_on = lane < WDQ * WROWS;
    // depthwise 1: (column x, channel quad): round rd covers x = 8 rd .. 8 rd + 7
    const int d1_q = lane & 7, d1_x0 = lane >> 3;
    // tap reads: conv pixel cr * CXW + 8 rd + d1_x0 + tap, chunk d1_q ^ pkey(pixel); (pixel & 7) = (d1_x0 + tap - cr) & 7, four cases
    unsigned d1f[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) d1f[v] = (unsigned)(OFF_CO + d1_x0 * COP * 4 + 16 * (d1_q ^ pkey((d1_x0 + v - 1) & 7)));
    // depthwise 2: (output column j, channel quad of 16): round rd covers j = 4 rd .. 4 rd + 3
    const int d2_q = lane & 15, d2_j = lane >> 4;
    // tap reads: pointwise pixel row * PXW + 8 rd + 2 d2_j + tap, chunk d2_q ^ pkey(pixel); (pixel & 7) = (5 row + 2 d2_j + tap) & 7, six cases
    // (2 d2_j is even: pixel 2 d2_j + 1 has the key of 2 d2_j with bit 2 set, its address is the same ^ 64; likewise row 1's taps 1
    // and 2 -- pixels 2 d2_j + 6 and + 7 -- so four of the six live in registers and two are one XOR away)
    static_assert(PXW % 8 == 5, "the key pairs of depthwise 2's taps");
    unsigned d2f[2][2];
    d2f[0][0] = (unsigned)(OFF_P1 + 2 * d2_j * P1P * 4 + 16 * (d2_q ^ pkey((2 * d2_j) & 7)));
    d2f[0][1] = (unsigned)(OFF_P1 + 2 * d2_j * P1P * 4 + 16 * (d2_q ^ pkey((2 * d2_j + 2) & 7)));
    d2f[1][0] = (unsigned)(OFF_P1 + 2 * d2_j * P1P * 4 + 16 * (d2_q ^ pkey((2 * d2_j + 5) & 7)));
    d2f[1][1] = (unsigned)(OFF_P1 + 2 * d2_j * P1P * 4 + 16 * (d2_q ^ pkey((2 * d2_j + 6) & 7)));

    typedef typename std::conditional<U8, unsigned, f32x4>::type raw_t;
    constexpr int VB = U8 ? 1 : 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
    u16x2 amax_pk = {0, 0};

    const unsigned nwaves = gridDim.x * WAVES;
    for (unsigned u = blockIdx.x * WAVES + wave; u < p.total; u += nwaves) {
        const unsigned lu = p.reverse ? p.total - 1u - u : u;
        const int seg = (int)(lu % (unsigned)p.segs);
        const unsigned t1 = lu / (unsigned)p.segs;
        const int strip = (int)(t1 % (unsigned)p.strips);
        const int n = (int)(t1 / (unsigned)p.strips);
        const int j0 = KS * strip, xb = 2 * j0, cb0 = xb - 1, wb = 2 * cb0;           // first output / block-1 / conv1 / input column
        const int i0 = seg * p.seg_rows, i1 = min(i0 + p.seg_rows, p.OH2);

        // column validity (SAME padding: a pixel outside its map must read as zero downstream) and, U8, the column half of the
        // mean case of a conv pixel
        float cval[CMB], pval[PMB];
        int ccase[CMB];
#pragma unroll
        for (int mb = 0; mb < CMB; ++mb) {
            const int cx = cb0 + c_rx[mb];
            cval[mb] = (cx >= 0 && cx < p.W1) ? 1.f : 0.f;
            ccase[mb] = cx == p.W1 - 1 ? 1 : 0;
        }
#pragma unroll
        for (int mb = 0; mb < PMB; ++mb) {
            const int px = xb + p_x[mb];
            pval[mb] = (px >= 0 && px < p.W1) ? 1.f : 0.f;
        }
        // window pieces of this lane: value offsets inside an input row, masked where the piece lies wholly outside the image
        const int q_hi = (3 * (p.W - wb) + WSHIFT) >> 2;
        const int q_lo = wb < 0 ? (3 * (-wb) + WSHIFT) >> 2 : 0;
        int wcol[2];
        bool wok[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = 2 * wl_dq + k;
            wcol[k] = wb * 3 - WSHIFT + 4 * q;
            wok[k] = wl_on && q >= q_lo && q < q_hi;
        }
        raw_t rawv[2];
        auto load_rows = [&](int s) __attribute__((always_inline)) {
            const int ih = 4 * s + 4 + wl_row;
            const bool rok = ih >= 0 && ih < p.H;
            const int rowpart = (n * p.H + ih) * p.W * 3;             // value index of the row's first value; < 2^29 (launcher)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const unsigned voff = (rok && wok[k]) ? (unsigned)((rowpart + wcol[k]) * VB) : 0x80000000u;
                if constexpr (U8) rawv[k] = __builtin_amdgcn_raw_buffer_load_b32(rx, voff, 0, S5_LDAUX);
                else rawv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, 0, S5_LDAUX));
            }
        };

        // depthwise accumulators: rows in flight.  dw1: per round, Q = the row that has its first kernel row, R is built fresh;
        // dw2: the output row that has its first kernel row
        f32x4 aQ[NR1], aR[NR1], aO[NR2];
#pragma unroll
        for (int r = 0; r < NR1; ++r) { aQ[r] = (f32x4){0.f, 0.f, 0.f, 0.f}; aR[r] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int r = 0; r < NR2; ++r) aO[r] = (f32x4){0.f, 0.f, 0.f, 0.f};

        load_rows(i0 - 2);
        for (int s = i0 - 2; s < i1; ++s) {
            // the constants a stage needs are READ WHERE THEY ARE USED: as loop invariants hipcc hoisted all of them (9 + 9 depthwise
            // taps, 14 scale / shift vectors: 130 registers) out of the step loop and spilled (DESIGN.md lesson 21) -- an opaque zero
            // in every address keeps the loads inside the step
            unsigned oz = 0;
            asm volatile("" : "+v"(oz));
            const float* const Kz = (const float*)((const unsigned char*)Kc + oz);
            const float* const Cz = (const float*)((const unsigned char*)Ct + oz);
            const float4* const W1z = (const float4*)((const unsigned char*)W1 + oz);
            const float4* const W2z = (const float4*)((const unsigned char*)W2 + oz);
            // ---- window: convert the five input rows once, park them as f16 planes ----
            if (wl_on) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    unsigned char* dst = L + wl_row * WRP + 8 * (2 * wl_dq + k);
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
                        const u32x2 hb = __builtin_bit_cast(u32x2, hi);
                        amax_pk = __builtin_elementwise_max(amax_pk, __builtin_bit_cast(u16x2, hb.x & 0x7FFF7FFFu));
                        amax_pk = __builtin_elementwise_max(amax_pk, __builtin_bit_cast(u16x2, hb.y & 0x7FFF7FFFu));
                    }
                }
            }
            wave_order();
            if (s + 1 < i1) load_rows(s + 1);        // in flight for the whole step

            // ---- conv1: the two new rows (2 s + 2, 2 s + 3) x CXW = 15 columns, straight from the window ----
            const int cr0 = 2 * s + 2;
            // rows of this step that lie outside their maps (SAME padding: they must read as zero downstream)
            {
                const float rv0 = (cr0 >= 0 && cr0 < p.H1) ? 1.f : 0.f, rv1 = (cr0 + 1 >= 0 && cr0 + 1 < p.H1) ? 1.f : 0.f;
                // the stage's constants once (LDS reads of constants were half of the kernel's LDS time: PMC, round 4)
                f32x4 cds[2], csh[2];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    cds[nb] = *(const f32x4*)(&Kz[nb * 16 + 4 * q4]);
                    if constexpr (!U8) csh[nb] = *(const f32x4*)(&Kz[32 + nb * 16 + 4 * q4]);
                }
#pragma unroll
                for (int mb = 0; mb < CMB; ++mb) {          // one row block at a time: 4 fragments + 2 accumulators live
                    f16x8 ah[2], al[2];
                    const unsigned char* a0 = L + caddr[mb];
                    ah[0] = ((const Frag4*)(a0))->v;
                    ah[1] = ((const Frag4*)(a0 + 16))->v;
                    if constexpr (!U8) { al[0] = ((const Frag4*)(a0 + WPLANE))->v; al[1] = ((const Frag4*)(a0 + WPLANE + 16))->v; }
                    const float vmul = cval[mb] * (c_ry[mb] ? rv1 : rv0);
                    f32x4 acc[2];
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int pdt = U8 ? 1 : 0; pdt < 3; ++pdt)
#pragma unroll
                            for (int nb = 0; nb < 2; ++nb)
                                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pdt == 1 ? cwl[st][nb] : cwh[st][nb], pdt == 0 ? al[st] : ah[st],
                                                                                 acc[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        f32x4 sh;
                        if constexpr (U8) {
                            const int cse = ((cr0 + c_ry[mb] == p.H1 - 1) ? 2 : 0) + ccase[mb];
                            sh = *(const f32x4*)(&Cz[cse * 32 + nb * 16 + 4 * q4]);
                        } else {
                            sh = csh[nb];
                        }
                        f32x4 o = vfma(acc[nb], cds[nb], sh);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = __builtin_amdgcn_fmed3f(o[e], 0.f, 6.f);
                        o = o * vmul;          // (a uniform `if` around it came back as 4 selects per vector: the multiply is cheaper)
                        *(f32x4*)(L + (coaddr[mb] ^ (64u * nb))) = o;
                    }
                    wave_order();
                }
            }
            wave_order();

            // ---- depthwise 1: each new conv row is read once and added into the three rows it belongs to ----
            {
                f32x4 w1[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) w1[k] = as_v(W1z[k * 8 + d1_q]);
                const f32x4 d1sc = *(const f32x4*)(&Kz[64 + 4 * d1_q]), d1sh = *(const f32x4*)(&Kz[96 + 4 * d1_q]);
#pragma unroll
                for (int cr = 0; cr < 2; ++cr) {
#pragma unroll
                    for (int rd = 0; rd < NR1; ++rd) {
                        // (lanes past the last column read on -- into cells of the A tile behind the conv rows, finite or not: their
                        // sums are never stored -- instead of repeating the last column: the address stays lane constant + immediate)
                        const int x = 8 * rd + d1_x0;
                        const f32x4 t0 = *(const f32x4*)(L + d1f[1 - cr] + (cr * CXW + 8 * rd) * COP * 4);
                        const f32x4 t1 = *(const f32x4*)(L + d1f[2 - cr] + (cr * CXW + 8 * rd + 1) * COP * 4);
                        const f32x4 t2 = *(const f32x4*)(L + d1f[3 - cr] + (cr * CXW + 8 * rd + 2) * COP * 4);
                        f32x4 sP = aQ[rd];                 // the row that had kernel rows 0 and 1: this is its third -> complete
                        sP = vfma(t0, w1[6], sP); sP = vfma(t1, w1[7], sP); sP = vfma(t2, w1[8], sP);
                        f32x4 sQ = aR[rd];                 // the row that had kernel row 0
                        sQ = vfma(t0, w1[3], sQ); sQ = vfma(t1, w1[4], sQ); sQ = vfma(t2, w1[5], sQ);
                        f32x4 sR = (f32x4){0.f, 0.f, 0.f, 0.f};
                        sR = vfma(t0, w1[0], sR); sR = vfma(t1, w1[1], sR); sR = vfma(t2, w1[2], sR);
                        aQ[rd] = sQ;
                        aR[rd] = sR;
                        // finished row -> scale, shift, ReLU6 (x 2^a), split -> A tile row cr * 9 + x
                        f32x4 v = vfma(sP, d1sc, d1sh);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, cap6);
                        const f16x4 hi = __builtin_convertvector(v, f16x4);
                        const f16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), f16x4);
                        const int q = cr * PXW + x;
                        if (8 * rd + 8 <= PXW || x < PXW) {
                            *(f16x4*)(L + OFF_AS + swzb(q, d1_q >> 1) + 8 * (d1_q & 1)) = hi;
                            *(f16x4*)(L + OFF_AS + swzb(q, 4 + (d1_q >> 1)) + 8 * (d1_q & 1)) = lo;
                        }
                        wave_order();
                    }
                    // (one conv row and one round at a time: with both in one scheduling region hipcc hoisted all twelve tap reads to the top and
                    // spilled lane constants -- whose reloads, vector-memory operations behind the row requests, then waited for HBM)
                    wave_order();
                }
            }
            wave_order();

            // ---- pointwise on the f16 MFMA: 18 pixels (two row blocks) x 64 channels, K = 32 in one instruction ----
            {
                // block-1 rows 2 s + 1 and 2 s + 2
                const int pr0 = 2 * s + 1;
                const float rv0 = (pr0 >= 0 && pr0 < p.H1) ? 1.f : 0.f, rv1 = (pr0 + 1 >= 0 && pr0 + 1 < p.H1) ? 1.f : 0.f;
                f16x8 bh[4], bl[4];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    bh[cb] = *(const f16x8*)(Wp + oz + (cb * 16 + l16) * 128 + 16 * (q4 ^ (l16 & 7)));
                    bl[cb] = *(const f16x8*)(Wp + oz + (cb * 16 + l16) * 128 + 16 * ((4 + q4) ^ (l16 & 7)));
                }
                f32x4 pds[4], psh[4];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) { pds[cb] = *(const f32x4*)(&Kz[128 + cb * 16 + 4 * q4]); psh[cb] = *(const f32x4*)(&Kz[192 + cb * 16 + 4 * q4]); }
#pragma unroll
                for (int mb = 0; mb < PMB; ++mb) {
                    const f16x8 ah = *(const f16x8*)(L + OFF_AS + swzb(16 * mb + l16, q4));
                    const f16x8 al = *(const f16x8*)(L + OFF_AS + swzb(16 * mb + l16, 4 + q4));
                    f32x4 acc[4];
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int pdt = 0; pdt < 3; ++pdt)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pdt == 1 ? bl[cb] : bh[cb], pdt == 0 ? al : ah, acc[cb], 0, 0, 0);
                    if (mb < PMB - 1 || p_store_last) {
                        const float vmul = pval[mb] * (p_pr[mb] ? rv1 : rv0);
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) {
                            f32x4 o = vfma(acc[cb], pds[cb], psh[cb]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = __builtin_amdgcn_fmed3f(o[e], 0.f, 6.f);
                            o = o * vmul;
                            *(f32x4*)(L + ((paddr[mb] + 64u * (cb & 2)) ^ (64u * (cb & 1)))) = o;
                        }
                    }
                    wave_order();
                }
            }
            wave_order();

            // ---- depthwise 2 (stride 2): output row s gets its second and third kernel rows, row s + 1 its first ----
            {
                const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + (size_t)n * p.OH2 * p.OW2 * 64, (long long)p.OH2 * p.OW2 * 256);
                const f32x4 d2sc = *(const f32x4*)(&Kz[256 + 4 * d2_q]), d2sh = *(const f32x4*)(&Kz[320 + 4 * d2_q]);
                f32x4 w2[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) w2[k] = as_v(W2z[k * 16 + d2_q]);
#pragma unroll
                for (int rd = 0; rd < NR2; ++rd) {
                    // (lanes past the last column read on, past the wave's own rows -- other waves' cells or the constants behind them, in
                    // bounds of the workgroup's LDS: never stored)
                    const f32x4 a0 = *(const f32x4*)(L + d2f[0][0] + (8 * rd) * P1P * 4), a1 = *(const f32x4*)(L + (d2f[0][0] ^ 64u) + (8 * rd + 1) * P1P * 4);
                    const f32x4 a2 = *(const f32x4*)(L + d2f[0][1] + (8 * rd + 2) * P1P * 4);
                    const f32x4 b0 = *(const f32x4*)(L + d2f[1][0] + (PXW + 8 * rd) * P1P * 4), b1 = *(const f32x4*)(L + d2f[1][1] + (PXW + 8 * rd + 1) * P1P * 4);
                    const f32x4 b2 = *(const f32x4*)(L + (d2f[1][1] ^ 64u) + (PXW + 8 * rd + 2) * P1P * 4);
                    f32x4 sO = aO[rd];
                    sO = vfma(a0, w2[3], sO); sO = vfma(a1, w2[4], sO); sO = vfma(a2, w2[5], sO);
                    sO = vfma(b0, w2[6], sO); sO = vfma(b1, w2[7], sO); sO = vfma(b2, w2[8], sO);
                    f32x4 sN = (f32x4){0.f, 0.f, 0.f, 0.f};
                    sN = vfma(b0, w2[0], sN); sN = vfma(b1, w2[1], sN); sN = vfma(b2, w2[2], sN);
                    aO[rd] = sN;
                    if (s >= i0) {
                        const f32x4 o = vfma(sO, d2sc, d2sh);
                        f32x4 v;
                        v[0] = apply_act<ACT>(o[0]); v[1] = apply_act<ACT>(o[1]); v[2] = apply_act<ACT>(o[2]); v[3] = apply_act<ACT>(o[3]);
                        const int ow = j0 + 4 * rd + d2_j;
                        // a column outside the strip or the map gets an offset beyond the resource and the store is dropped (no branch)
                        const unsigned voff = (4 * rd + d2_j < KS && ow < p.OW2) ? (unsigned)(s * p.OW2 + ow) * 256u + 16u * d2_q : 0x80000000u;
                        bstore16(v, ry, voff, 0);
                    }
                    wave_order();          // (one round at a time, as in depthwise 1)
                }
            }
            wave_order();
        }
    }
    if constexpr (!U8) {
        const unsigned am = max((unsigned)amax_pk[0], (unsigned)amax_pk[1]);
        if (am >= 0x7800u && p.overflow) atomicOr(p.overflow, 1);
    }
}

}  // namespace

HSEFR_KNOB(g_stem5_grid, 256 * S5_WGS);   // dev builds: workgroups of the launch (S5_WGS per CU resident)
HSEFR_KNOB(g_stem5_segs, 0);        // dev builds: vertical segments per strip (0 = chosen by the launcher)
#ifdef HSEFR_DEV
void set_stem5_grid(int v) { g_stem5_grid = v; }
void set_stem5_segs(int v) { g_stem5_segs = v; }
#endif

bool stem5_stream_supported(int cin, int c1, int c2, int conv_stride, int dw1_stride, int dw2_stride, int kh, int kw, int h, int w) {
    return cin == 3 && c1 == 32 && c2 == 64 && conv_stride == 2 && dw1_stride == 1 && dw2_stride == 2 && kh == 3 && kw == 3 &&
           h >= 4 && w >= 4 && h % 4 == 0 && w % 4 == 0;
}

int launch_stem5_stream(const void* x, int x_is_u8, const void* cw4, const float* cdescale, const float* cshift, const float* wd1,
                        const float* d1scale, const float* d1shift, const void* wsplit, const float* descale, const float* pshift,
                        const float* wd2, const float* d2scale, const float* d2shift, float* y, int* overflow, int n, int h, int w,
                        int in_log2, int a_log2, int act, hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && h >= 4 && w >= 4 && h % 4 == 0 && w % 4 == 0, HSEFR_ERR_INVALID,
                  "stem5_stream: %dx%d input (both edges must be multiples of 4; other sizes take stem3_fused)", h, w);
    HSEFR_REQUIRE(a_log2 > 0 && a_log2 <= 12, HSEFR_ERR_INVALID, "stem5_stream: a_log2=%d", a_log2);
    HSEFR_REQUIRE(x_is_u8 ? in_log2 == 0 : (in_log2 >= -8 && in_log2 <= 14), HSEFR_ERR_INVALID, "stem5_stream: in_log2=%d", in_log2);
    // input offsets travel in 32 bits and 2^31 marks a masked piece: a launch takes at most 2 GB of (fp32-sized) input -- 4 850
    // images of 192 x 192; a larger batch goes as several launches over ranges of images
    const long long per_img = (long long)h * w * 12;
    HSEFR_REQUIRE(per_img < (1ll << 31) - 64, HSEFR_ERR_UNSUPPORTED, "stem5_stream: one %dx%d image exceeds the 2 GB a launch addresses", h, w);
    // the OUTPUT is addressed through one buffer resource per image (OH2 x OW2 x 256 B = h w 16 B) with the same 2^31 marker for
    // masked stores: an image whose map reaches 2 GB would put the marker INSIDE the resource
    HSEFR_REQUIRE((long long)h * w * 16 < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "stem5_stream: the %dx%d image's output map exceeds 2 GB", h, w);
    if (n == 0) return HSEFR_OK;
    const long long n_max = ((1ll << 31) - 64 - 1) / per_img;
    if (n > n_max) {
        for (long long i0 = 0; i0 < n; i0 += n_max) {
            const int m = (int)(n - i0 < n_max ? n - i0 : n_max);
            const int rc = launch_stem5_stream((const char*)x + i0 * h * w * 3 * (x_is_u8 ? 1 : 4), x_is_u8, cw4, cdescale, cshift, wd1, d1scale, d1shift,
                                               wsplit, descale, pshift, wd2, d2scale, d2shift, y + i0 * (h / 4) * (w / 4) * 64, overflow, m, h, w,
                                               in_log2, a_log2, act, s);
            if (rc != HSEFR_OK) return rc;
        }
        return HSEFR_OK;
    }
    Stem5Params p;
    p.x = x; p.cw4 = cw4; p.cdescale = cdescale; p.cshift = cshift; p.wd1 = (const float4*)wd1; p.d1scale = (const float4*)d1scale;
    p.d1shift = (const float4*)d1shift; p.wsplit = (const float*)wsplit; p.descale = descale; p.pshift = pshift;
    p.wd2 = (const float4*)wd2; p.d2scale = (const float4*)d2scale; p.d2shift = (const float4*)d2shift; p.y = y; p.overflow = overflow;
    p.H = h; p.W = w; p.H1 = h / 2; p.W1 = w / 2; p.OH2 = h / 4; p.OW2 = w / 4;
    p.strips = (p.OW2 + KS - 1) / KS;
    // vertical segments: a strip costs (rows + 2) steps (two start-up steps whose output is dropped); waves take units round robin
    // -- choose the split that minimises rounds x steps per unit over the resident wave slots
    const long long slots = (long long)g_stem5_grid * WAVES;
    const long long strips_total = (long long)n * p.strips;
    int best = 1;
    long long best_cost = -1;
    for (int sg = 1; sg <= p.OH2; ++sg) {
        const int rows = (p.OH2 + sg - 1) / sg;
        if ((long long)rows * (sg - 1) >= p.OH2) continue;             // the last segment would be empty
        const long long rounds = (strips_total * sg + slots - 1) / slots;
        const long long cost = rounds * (rows + 2);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sg; }
    }
    if (g_stem5_segs > 0 && g_stem5_segs <= p.OH2 && (long long)((p.OH2 + g_stem5_segs - 1) / g_stem5_segs) * (g_stem5_segs - 1) < p.OH2)
        best = g_stem5_segs;
    p.segs = best;
    p.seg_rows = (p.OH2 + best - 1) / best;
    const long long total = strips_total * best;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "stem5_stream: grid too large");
    p.total = (unsigned)total;
    p.a_scale = ldexpf(1.f, a_log2);
    p.in_scale = ldexpf(1.f, in_log2);
    p.reverse = sweep_reverse();
    p.x_bytes = (long long)n * h * w * 3 * (x_is_u8 ? 1 : 4);
    const long long wgs = (total + WAVES - 1) / WAVES;
    const unsigned g = (unsigned)(wgs < g_stem5_grid ? wgs : g_stem5_grid);
#define HSEFR_STEM5(A)                                                                                            \
    do {                                                                                                          \
        if (x_is_u8) HSEFR_LAUNCH((stem5_stream_kernel<A, true>), dim3(g), dim3(64 * WAVES), 0, s, p);      \
        else HSEFR_LAUNCH((stem5_stream_kernel<A, false>), dim3(g), dim3(64 * WAVES), 0, s, p);             \
    } while (0)
    if (act == HSEFR_ACT_RELU6) HSEFR_STEM5(HSEFR_ACT_RELU6);
    else if (act == HSEFR_ACT_RELU) HSEFR_STEM5(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_NONE) HSEFR_STEM5(HSEFR_ACT_NONE);
    else { set_error("stem5_stream: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_STEM5
    return launch_status("stem5_stream");
}

}  // namespace hsefr
