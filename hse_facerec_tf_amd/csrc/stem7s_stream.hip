// ResNet stem, STREAMING form (round 6): 7x7 / stride 2 / pad 3 convolution over the 3-channel fp32 image + scale + shift + ReLU
// -> 3x3 / stride 2 max-pool -> bf16 NHWC (conv1/7x7_s2 + BN + ReLU + pool1/3x3_s2 of resnet50_ft, the graph behind vgg2_resnet.pb at
// facerec_test.py:213), gfx950.  Same operation and rounding points as stem7x7_pool.hip (bf16 input, exact products, fp32
// accumulation, bf16 after the ReLU; the max-pool commutes with the rounding), another decomposition -- MobileNet's stem went the same
// way in round 4 (stem5_stream.hip, DESIGN.md lesson 39):
//
//   * stem7x7_pool.hip gives a workgroup a 7 x 7 tile of pooled pixels and walks it through three barrier-separated phases (window
//     scatter, implicit GEMM with the weights in 88 registers per lane, pooling out of a 32 KB conv tile): 249 registers, two workgroups
//     per CU, 12 k cycles per tile of which 1.4 k are MFMA, 41 % of its LDS cycles bank conflicts (2-byte window stores, 8-byte conv-tile
//     stores, 4-byte-aligned fragment reads): 93 us at batch 128 against an HBM floor of 20.
//   * Here ONE WAVE = ONE STRIP of KS = 7 pooled columns, walking DOWN the image one pooled row per step.  A step brings in four new
//     input rows, computes the TWO new conv rows (15 columns each: one MFMA row block per row) and keeps the horizontal maxima of the
//     previous step's last conv row in registers: the conv map never exists, there is no vertical halo, no barrier after the prologue,
//     and the waves of a CU drift apart so that the matrix pipe, the vector ALU and the LDS work for different waves at once.
//   * The window lives in LDS as [row][pixel][R, G, B, 0] bf16 -- EIGHT bytes per pixel: kernel row dy of conv column cx is the 32
//     values from pixel 2 cx on (7 pixels x 4 + one pixel that meets zero weights), so K = 7 x 32 and every window fragment is ONE 16-byte
//     ALIGNED ds_read_b128 (the 6-byte pixels of the patch kernel put fragments at 4-byte alignment: two ds_read2_b32 each); a pixel is
//     one buffer_load_dwordx3, masked as a whole where it lies outside the image (out-of-range offset: zeros); a 12-row ring, written
//     at the top of the step the rows are first needed in (requested a step earlier), three step phases with immediate slot offsets.
//   * Pixels are the MFMA's ROWS (A = window fragments, B = weights): a lane ends with conv columns 4 q4 + e of ONE channel per 16-channel
//     block, so scale and shift are two registers per block.  (With the channels as rows -- the first form -- they were 16-byte LDS reads
//     per block and step: four exposed LDS round trips, half of a step's cycles by the stamps.)
//   * The pool's maxima commute: the three ROWS first, in the lane (the previous step's second conv row is carried, ReLU rides in it), then
//     the row of column-wise maxima goes to LDS as bf16 (rounding is monotonic: once, before the last maximum) and leaves as the maximum
//     of conv columns 2 j, 2 j + 1, 2 j + 2 -- v_pk_max_u16 on non-negative bf16 bits -- in 16-byte stores.
//   * Rows / columns outside the conv map count as 0 under a uniform branch (border steps only).
// Weights (7 x 4 fragments per lane, 112 registers): the workgroup re-orders the blob's [64][8][32] image (k = dy * 32 + dx * 3 + ci) ONCE
// into fragment order in LDS, every lane reads 28 x 16 bytes (each wave gathering its own 224 halfwords was a quarter of its lifetime).
// Measured (batch 128, 224 x 224, in the network): 95.0 us (the patch kernel) -> 63.8 us.  tools/s7_stamps.py (-DHSEFR_S7_STAMPS) prints
// where a wave's cycles go; the S7_KO builds (tools/build_ko.sh) knock one part out at a time.
#include <type_traits>

#include "common.h"

#ifndef S7_LDAUX
#define S7_LDAUX 2   // cache policy of the image loads (buffer aux bits: 1 glc, 2 slc): streamed once -- with slc the pooled map survives for its two readers (the projection pair behind the stem 98 -> 90 us, the stem itself + 1.5)
#endif
#ifndef S7_KO
#define S7_KO 0      // knock-out builds (timing only, results WRONG): 1 = no MFMAs, 2 = no scale / ReLU / horizontal maxima, 4 = the window loads move no bytes, 8 = no output stores
#endif
namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned short u16;

struct Stem7sParams {
    const float* x;      // [N,H,W,3] fp32
    const u16* wt;       // [64][8][32] bf16, k = dy*32 + dx*3 + ci, zero padded (resnet50.pack_stem_weight)
    const bf16x8* wfrag; // the same weights in fragment order (stem7s_reorder_kernel), or null: the kernel re-orders wt itself
    const float* scale;  // [64]
    const float* shift;  // [64]
    u16* y;              // [N,PH,PW,64] bf16
    int H, W, OH, OW, PH, PW, ppt, ppl;
    int strips, segs, seg_rows;     // strips per pooled row, vertical segments per strip, pooled rows per segment
    unsigned total;                 // units = N * strips * segs
    long long x_bytes, y_bytes;
    int reverse;
};

constexpr int KS = 7;                    // pooled columns of a strip
constexpr int CXW = 2 * KS + 1;          // conv columns: 15 (one MFMA row block per conv row; lane 15 repeats column 14)
constexpr int WPX = 2 * (CXW - 1) + 7;   // input pixels of a window row: 35
constexpr int WROWB = (2 * (CXW - 1) + 8) * 8;   // bytes per window row: the last fragment's eighth pixel is read (against zero weights): 36 pixels
constexpr int RING = 12;                 // window rows in LDS: nine live + the four a step replaces, as three phases of four
constexpr int STG_OFF = RING * WROWB;    // 3456: the step's row of column-wise maxima on its way out: 16 conv columns x 64 channels bf16
constexpr int STGPX = 144;               // bytes per staged conv column: 128 + 16 (the four 4-column lane groups of a write land on different banks)
constexpr int WAVE_LDS = (STG_OFF + 16 * STGPX + 127) / 128 * 128;      // 5760
constexpr int WAVES = 4;
constexpr int NEWPX = 4 * WPX;           // pixels of a step's four new rows: 140 -> three rounds of 64 lanes
static_assert(WROWB % 16 == 0 && STG_OFF % 16 == 0, "16-byte aligned fragments and stores");

__device__ __forceinline__ void wave_order() { asm volatile("" ::: "memory"); }

#ifdef HSEFR_S7_STAMPS      // development: where a wave's cycles go, per step phase (tools/s7_stamps.py)
__device__ unsigned long long g_s7_stamps[512 * 4 * 10];
#define S7_STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); st[i] += _t - tprev; tprev = _t; } while (0)
#define S7_STAMP_DECL unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tprev = __builtin_amdgcn_s_memtime(); const unsigned long long tstart = tprev; unsigned long long nst = 0
#define S7_STAMP_RESET do { tprev = __builtin_amdgcn_s_memtime(); } while (0)
#define S7_STAMP_FLUSH do { if (lane == 0 && blockIdx.x < 512) { unsigned long long* o = g_s7_stamps + (blockIdx.x * 4 + wave) * 10; \
    for (int i_ = 0; i_ < 8; ++i_) o[i_] = st[i_]; o[8] = __builtin_amdgcn_s_memtime() - tstart; o[9] = nst; } } while (0)
#else
#define S7_STAMP(i) do { } while (0)
#define S7_STAMP_DECL do { } while (0)
#define S7_STAMP_RESET do { } while (0)
#define S7_STAMP_FLUSH do { } while (0)
#endif
// fragment f = (nb * 7 + dy) * 64 + lane of the weight image: channel 16 nb + lane % 16, kernel row dy, pixels 2 (lane / 16) and + 1 as
// [R, G, B, 0] each, from the blob's [64][8][32] image (k = dy * 32 + dx * 3 + ci)
template <typename P> __device__ __forceinline__ bf16x8 wfrag_of(P img, int f) {
    const int fl = f & 63, fdy = (f >> 6) % 7, fnb = f / (64 * 7);
    const auto row = img + (16 * fnb + (fl & 15)) * 256 + fdy * 32;
    unsigned short v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int dx = 2 * (fl >> 4) + (i >> 2), ci = i & 3;
        v[i] = (dx < 7 && ci < 3) ? row[dx * 3 + ci] : (unsigned short)0;
    }
    return __builtin_bit_cast(bf16x8, *(const __attribute__((ext_vector_type(8))) unsigned short*)v);
}

// the engine's one-off: the whole image in fragment order (STEM7S_WFRAG_BYTES)
__global__ __launch_bounds__(256) void stem7s_reorder_kernel(const u16* wt, bf16x8* out) {
    for (int f = threadIdx.x; f < 4 * 7 * 64; f += 256) out[f] = wfrag_of(wt, f);
}

__global__ __launch_bounds__(64 * WAVES, 2) void stem7s_stream_kernel(Stem7sParams p) {
    __shared__ __attribute__((aligned(128))) unsigned char smem[64 * 512 + 4 * 7 * 64 * 16];       // the weight image (32 KB) + its fragment-ordered copy (28 KB) during the prologue, then WAVES x WAVE_LDS
    static_assert(WAVES * WAVE_LDS <= 64 * 512, "the wave regions fit where the weight image was");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, q4 = lane >> 4;
    S7_STAMP_DECL;

    // ---- prologue: the weight image into fragment registers, re-ordered to four values per pixel.  With the engine's fragment-ordered copy
    // (p.wfrag: made once per plan) that is 28 16-byte loads per lane; without it (direct calls) the workgroup re-orders the blob's image
    // through LDS -- [nb][dy][lane] x 16 bytes behind the image, every lane then reads 28 ds_read_b128 (each wave gathering its own
    // registers was 224 ds_read_u16 per lane: a quarter of a wave's lifetime; the workgroup's pass is still a tenth) ----
    bf16x8 wf[4][7];     // [channel block nb][kernel row dy]: lane (channel 16 nb + l16, k slice q4) holds k = 32 dy + 8 q4 .. + 7 = pixels 2 q4, 2 q4 + 1
    float scl[4], shf[4];      // the lane's channel of every block (16 nb + l16): scale and shift stay in registers
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) { scl[nb] = p.scale[16 * nb + l16]; shf[nb] = p.shift[16 * nb + l16]; }
    if (p.wfrag) {       // (uniform)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int dy = 0; dy < 7; ++dy) wf[nb][dy] = p.wfrag[(nb * 7 + dy) * 64 + lane];
    } else {
        for (int i = tid; i < 64 * 512 / 16; i += 64 * WAVES) ((f32x4*)smem)[i] = ((const f32x4*)p.wt)[i];
        __syncthreads();
        for (int f = tid; f < 4 * 7 * 64; f += 64 * WAVES) *(bf16x8*)(smem + 64 * 512 + f * 16) = wfrag_of((const u16*)smem, f);
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int dy = 0; dy < 7; ++dy) wf[nb][dy] = *(const bf16x8*)(smem + 64 * 512 + ((nb * 7 + dy) * 64 + lane) * 16);
        __syncthreads();     // the image's bytes become the waves' regions
    }
    unsigned char* const L = smem + wave * WAVE_LDS;
    S7_STAMP(7);         // the prologue: weights into fragment registers
    // the eighth pixel of a row's last fragment (pixel 35) is written by no load: zero once (finite under its zero weights)
    if (lane < RING) *(u32x2*)(L + lane * WROWB + WPX * 8) = u32x2{0u, 0u};

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
    const int l16c = l16 < CXW ? l16 : CXW - 1;
    const unsigned fr0 = (unsigned)(16 * (l16c + q4));          // byte offset of the lane's B fragment inside a window row: pixels 2 cx + 2 q4, + 1

    // lane roles of the window rounds (the same for every unit): slot i = lane + 64 r -> (new row i / 35, pixel i % 35)
    int wrow[3], wpx[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int i = lane + 64 * r;
        wrow[r] = i < NEWPX ? i / WPX : -1;
        wpx[r] = i - WPX * (i / WPX);
    }

    const unsigned nwaves = gridDim.x * WAVES;
    for (unsigned u = blockIdx.x * WAVES + wave; u < p.total; u += nwaves) {
        const unsigned lu = p.reverse ? p.total - 1u - u : u;
        const int seg = (int)(lu % (unsigned)p.segs);
        const unsigned t1 = lu / (unsigned)p.segs;
        const int strip = (int)(t1 % (unsigned)p.strips);
        const int n = (int)(t1 / (unsigned)p.strips);
        const int px0 = KS * strip, cx0 = 2 * px0 - p.ppl, ix0 = 2 * cx0 - 3;
        const int py_a = seg * p.seg_rows, py_b = min(py_a + p.seg_rows, p.PH);
        const int nsteps = py_b - py_a + 1;                       // one start-up step (its pooled row is dropped) + one per pooled row
        const int cA0 = 2 * py_a - p.ppt - 1;                     // first conv row of the start-up step
        const int Rb = 2 * cA0 - 3;                               // its first input row: window row r of the unit is input row Rb + r

        // per-lane constants of the unit: column validity of the lane's conv column and of its window pixels, offsets of the latter
        bool cval[4];        // the lane's conv columns 4 q4 + e inside the map (the strip's sixteenth column repeats the fifteenth and is read by no pooled one)
#pragma unroll
        for (int e = 0; e < 4; ++e) cval[e] = cx0 + 4 * q4 + e >= 0 && cx0 + 4 * q4 + e < p.OW;
        unsigned woff[3];
        bool wcol[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ix = ix0 + wpx[r];
            wcol[r] = wrow[r] >= 0 && ix >= 0 && ix < p.W;
            woff[r] = (unsigned)(((n * p.H + Rb + 5 + wrow[r]) * p.W + ix) * 12);      // row 5 + wrow of the unit; + 4 rows per step
        }
        const unsigned rowstep = (unsigned)(4 * p.W * 12);
        u32x3 raw[3];
        auto load_new = [&](int s) __attribute__((always_inline)) {      // the four rows step s writes: unit rows 4 s + 5 .. 4 s + 8
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = Rb + 4 * s + 5 + wrow[r];
                const bool ok = wcol[r] && iy >= 0 && iy < p.H;
                raw[r] = __builtin_amdgcn_raw_buffer_load_b96(rx, (ok && !(S7_KO & 4)) ? woff[r] + (unsigned)s * rowstep : 0x80000000u, 0, S7_LDAUX);
            }
        };
        auto to_px = [](u32x3 v) __attribute__((always_inline)) -> u32x2 {
            return u32x2{hsefr_pack_bf16x2(__uint_as_float(v.x), __uint_as_float(v.y)), hsefr_pack_bf16x2(__uint_as_float(v.z), 0.f)};
        };
        // the five rows the first step has from "before": unit rows 0 .. 4 (175 pixels, three rounds), straight into their slots
        {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int i = lane + 64 * r;
                const int rr = i / WPX, pp = i - WPX * rr;
                const int iy = Rb + rr, ix = ix0 + pp;
                const bool ok = rr < 5 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rx, ok ? (unsigned)(((n * p.H + iy) * p.W + ix) * 12) : 0x80000000u, 0, S7_LDAUX);
                if (rr < 5) *(u32x2*)(L + rr * WROWB + pp * 8) = to_px(v);
            }
        }
        load_new(0);
        S7_STAMP(7);             // (unit set-up: the five rows from "before" count with the prologue)

        f32x4 carry[4];      // the previous step's second conv row after scale, shift and ReLU (the next pooled row's first)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) carry[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto step = [&](auto PHASE, int s) __attribute__((always_inline)) {
            constexpr int PH3 = decltype(PHASE)::value;          // s % 3: window row i of the step sits in ring slot (4 PH3 + i) % 12
            // ---- the four new rows (requested a step ago) into their slots; the next step's rows requested ----
            // (no branch around a store: behind one hipcc waits for EVERY outstanding vector-memory operation -- the previous step's
            // output store included, a write acknowledgement per step -- instead of counting the loads; a lane without a pixel in the
            // third round loaded zeros and writes them onto row 0's spare pixel, which is zero anyway)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                int slot = 4 * PH3 + 5 + wrow[r];
                slot = slot >= RING ? slot - RING : slot;
                const unsigned dst = wrow[r] >= 0 ? (unsigned)(slot * WROWB + wpx[r] * 8) : (unsigned)(WPX * 8);
                *(u32x2*)(L + dst) = to_px(raw[r]);
            }
            wave_order();
            S7_STAMP(0);         // the window rows: wait for the loads of a step ago, convert, write
            if (s + 1 < nsteps) load_new(s + 1);
            S7_STAMP(1);         // the next step's loads requested
            // ---- the two new conv rows: implicit GEMM straight off the window, 32 output channels at a time (two passes over the
            // fourteen fragments: sixteen accumulator registers instead of thirty-two -- with 112 registers of weights the full set spilled) ----
            const int cA = cA0 + 2 * s;
            const int py = py_a + s - 1;
            const bool edge = cA < 0 || cA + 1 >= p.OH || cx0 < 0 || cx0 + 16 > p.OW;      // (uniform) some conv row / column of the step lies outside the map
            const bool rval[2] = {cA >= 0 && cA < p.OH, cA + 1 >= 0 && cA + 1 < p.OH};
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                f32x4 acc[2][2];
#pragma unroll
                for (int cr = 0; cr < 2; ++cr)
#pragma unroll
                    for (int k = 0; k < 2; ++k) acc[cr][k] = f32x4{0.f, 0.f, 0.f, 0.f};
                // the step's NINE window rows at the lane's column: conv row cr, kernel row dy is window row 2 cr + dy -- rows 2 .. 6 serve both
                // conv rows (9 ds_read_b128 per half where "7 per conv row" was 14; kept across both channel halves they spill)
                bf16x8 xr[9];
#pragma unroll
                for (int r = 0; r < 9; ++r) xr[r] = *(const bf16x8*)(L + ((4 * PH3 + r) % RING) * WROWB + fr0);
#pragma unroll
                for (int dy = 0; dy < 7; ++dy)
#pragma unroll
                    for (int cr = 0; cr < 2; ++cr) {
#if !(S7_KO & 1)
#pragma unroll
                        for (int k = 0; k < 2; ++k)
                            acc[cr][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xr[2 * cr + dy], wf[2 * hb + k][dy], acc[cr][k], 0, 0, 0);
#else
                        acc[cr][0][0] += (float)xr[2 * cr + dy][0];       // (knock-out build: the fragment stays "used")
#endif
                    }
                S7_STAMP(2 + 2 * hb);      // fragment reads + MFMAs of the half
                // ---- scale, shift, ReLU and the pool's three ROWS, in the lane (pixels are the MFMA's rows here: the lane holds conv columns
                // 4 q4 + e of ONE channel per block, so scale and shift are two registers per block -- with the channels as rows they were
                // 16-byte LDS reads per block and step, four exposed round trips).  Maxima commute: rows first, the columns when the row
                // leaves.  Rows / columns outside the conv map count as 0 (the clipped window's maximum is >= 0 after ReLU): masked under a
                // uniform branch, the image's border steps only.  Rounding to bf16 happens once per column-wise maximum: rounding is
                // monotonic, so the maximum of rounded values is the rounded maximum. ----
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int nb = 2 * hb + k;
                    f32x4 v[2];
#pragma unroll
                    for (int cr = 0; cr < 2; ++cr)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#if !(S7_KO & 2)
                            v[cr][e] = fmaf(acc[cr][k][e], scl[nb], shf[nb]);
#else
                            v[cr][e] = acc[cr][k][e];
#endif
                    if (edge) {
                        wave_order();      // (an asm statement is not speculated: the branch stays a branch instead of 32 selects on every step)
#pragma unroll
                        for (int cr = 0; cr < 2; ++cr)
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[cr][e] = (rval[cr] && cval[e]) ? v[cr][e] : 0.f;
                    }
                    f32x4 t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t[e] = fmaxf(fmaxf(carry[nb][e], v[0][e]), v[1][e]);      // carry >= 0: ReLU rides in it
                        carry[nb][e] = fmaxf(v[1][e], 0.f);
                    }
                    // (the start-up step's row is computed and dropped: no branch -- see the window stores)
                    const unsigned p01 = hsefr_pack_bf16x2(t[0], t[1]), p23 = hsefr_pack_bf16x2(t[2], t[3]);
                    unsigned char* const so = L + STG_OFF + (4 * q4) * STGPX + (16 * nb + l16) * 2;
                    *(u16*)(so) = (u16)p01; *(u16*)(so + STGPX) = (u16)(p01 >> 16);
                    *(u16*)(so + 2 * STGPX) = (u16)p23; *(u16*)(so + 3 * STGPX) = (u16)(p23 >> 16);
                }
                S7_STAMP(3 + 2 * hb);      // the half's epilogue
            }
            {
                wave_order();
                // lanes 0 .. 8 KS - 1: pooled column j = lane / 8, 16-byte chunk lane % 8 of its 128 bytes = the maximum of conv columns 2 j,
                // 2 j + 1, 2 j + 2.  The values are >= 0: bf16 bit patterns order like unsigned integers (v_pk_max_u16).
                const int j = lane >> 3;
                const unsigned char* const si = L + STG_OFF + (lane < 8 * KS ? 2 * j * STGPX + (lane & 7) * 16 : 0);
                typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
                const u16x8 c0 = *(const u16x8*)si, c1 = *(const u16x8*)(si + STGPX), c2 = *(const u16x8*)(si + 2 * STGPX);
                const u16x8 mx = __builtin_elementwise_max(__builtin_elementwise_max(c0, c1), c2);
                const f32x4 o = __builtin_bit_cast(f32x4, mx);
                const bool ok = s >= 1 && lane < 8 * KS && px0 + j < p.PW && !(S7_KO & 8);
                const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y, p.y_bytes);
                bstore16(o, ry, ok ? (unsigned)(((n * p.PH + py) * p.PW + px0 + j) * 128 + (lane & 7) * 16) : 0x80000000u, 0);
                wave_order();
                S7_STAMP(6);               // the pooled row: staging -> 16-byte stores
            }
#ifdef HSEFR_S7_STAMPS
            ++nst;
#endif
        };
        for (int s = 0; s < nsteps; s += 3) {
            step(std::integral_constant<int, 0>(), s);
            if (s + 1 < nsteps) step(std::integral_constant<int, 1>(), s + 1);
            if (s + 2 < nsteps) step(std::integral_constant<int, 2>(), s + 2);
        }
        wave_order();
    }
    S7_STAMP_FLUSH;
}

HSEFR_KNOB(g_stem7s, 1);     // dev builds: 0 = the patch kernel (stem7x7_pool.hip) also where this one covers the shape (A/B timing)

}  // namespace

#ifdef HSEFR_DEV
void set_stem7s(int v) { g_stem7s = v; }
int read_s7_stamps(void* host_out, size_t bytes) {
#ifdef HSEFR_S7_STAMPS
    HSEFR_REQUIRE(bytes <= sizeof(unsigned long long) * 512 * 4 * 10, HSEFR_ERR_INVALID, "read_s7_stamps: too many bytes");
    HSEFR_HIP_CHECK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_s7_stamps), bytes));
    return HSEFR_OK;
#else
    (void)host_out; (void)bytes;
    set_error("read_s7_stamps: library built without -DHSEFR_S7_STAMPS");
    return HSEFR_ERR_UNSUPPORTED;
#endif
}
#endif

bool stem7s_stream_supported(long long n, int h, int w, int ph, int pw) {
    const long long oh = (h - 1) / 2 + 1, ow = (w - 1) / 2 + 1;
    return g_stem7s != 0 && n > 0 && h >= 8 && w >= 8 && ph >= 1 && pw >= 1 && n * h * w * 12 < (1ll << 31) && n * ph * pw * 128 < (1ll << 31) && oh > 0 && ow > 0;
}

int launch_stem7s_stream(const float* x, const void* wt, const float* scale, const float* shift, void* y, int n, int h, int w, int ph, int pw,
                         int pool_pad_t, int pool_pad_l, hipStream_t s, const void* wfrag) {
    HSEFR_REQUIRE(stem7s_stream_supported(n, h, w, ph, pw), HSEFR_ERR_UNSUPPORTED, "stem7s_stream: shape not covered");
    Stem7sParams p;
    p.x = x; p.wt = (const u16*)wt; p.wfrag = (const bf16x8*)wfrag; p.scale = scale; p.shift = shift; p.y = (u16*)y;
    p.H = h; p.W = w; p.OH = (h - 1) / 2 + 1; p.OW = (w - 1) / 2 + 1; p.PH = ph; p.PW = pw; p.ppt = pool_pad_t; p.ppl = pool_pad_l;
    p.strips = (pw + KS - 1) / KS;
    // vertical segments: enough units for the 2048 resident waves (a segment costs one start-up step)
    const long long strips_all = (long long)n * p.strips;
    int segs = (int)((2048 + strips_all - 1) / strips_all);
    if (segs > ph / 4) segs = ph / 4;
    if (segs < 1) segs = 1;
    p.seg_rows = (ph + segs - 1) / segs;
    p.segs = (ph + p.seg_rows - 1) / p.seg_rows;
    p.total = (unsigned)(strips_all * p.segs);
    p.x_bytes = (long long)n * h * w * 12;
    p.y_bytes = (long long)n * ph * pw * 128;
    p.reverse = sweep_reverse();
    const unsigned need = (p.total + WAVES - 1) / WAVES;
    HSEFR_LAUNCH(stem7s_stream_kernel, dim3(need < 512u ? need : 512u), dim3(64 * WAVES), 0, s, p);
    return launch_status("stem7s_stream");
}

int stem7s_reorder_weights(const void* wt, void* wfrag, hipStream_t s) {
    HSEFR_LAUNCH(stem7s_reorder_kernel, dim3(1), dim3(256), 0, s, (const u16*)wt, (bf16x8*)wfrag);
    return launch_status("stem7s_reorder");
}

}  // namespace hsefr
