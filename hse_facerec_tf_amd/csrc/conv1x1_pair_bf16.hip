// Two chained 1x1 bf16 convolutions in ONE launch (round 6): a bottleneck's "increase" layer (+ residual | + projected shortcut,
// + ReLU) and the NEXT bottleneck's "reduce" layer (+ ReLU) of ResNet-50 (resnet50_ft, the graph behind vgg2_resnet.pb at
// facerec_test.py:213).  NHWC bf16 in / out, fp32 accumulation, gfx950; rounding points of oracle/resnet50.py, the same as the two
// launches this replaces:
//
//   Y1[p, n] = act1( bf16( s1[n] * sum_k X[p, k] * W1[n, k] + b1[n] ) + R[p, n] )                  (stored: the next block's shortcut)
//   Y2[p, m] = act2( bf16( s2[m] * sum_n Y1[p, n] * W2[m, n] + b2[m] ) )                           (stored: the next 3x3's input)
//   PROJ:  R[p, n] = bf16( sp[n] * sum_k X2[p, k] * WP[n, k] + bp[n] )      (the block input's projection, never stored)
//
// Why: in the 56-pixel stage both layers run at HBM speed (profiles/r05_resnet50_layers.txt: 5.3-5.5 TB/s), and the reduce layer's only
// HBM read is the tensor the increase layer has just written -- 205 MB per pair at batch 128, 40 % of what the pair moves.  Here a
// wave keeps its pixels' Y1 in registers:
//   * weights first in v_mfma_f32_16x16x32_bf16 with permuted weight rows (conv1x1_w4_bf16.hip): a lane leaves a pair of 16-channel
//     blocks with EIGHT CONSECUTIVE channels of one pixel -- which is (a) a 16-byte store of Y1 and (b), rounded to bf16, exactly the
//     B operand of the second product's 32-channel K chunk (lane (pixel l16, k group lq) holds k = 8 lq .. 8 lq + 7): the chain needs no
//     LDS transpose, no shuffle, and W2 keeps its natural K order;
//   * the walk is CHUNKED over Y1's channels: 32 channels at a time -- first product (K1 / 32 x 2 MFMAs per pixel block), epilogue,
//     store, then that chunk's contribution to every channel of Y2 -- so only 2 x PB accumulator blocks of Y1 exist at any time;
//   * all three weight matrices are RESIDENT in LDS (32 KB each at 64 -> 256 -> 64), loaded once per workgroup; the waves share nothing
//     else: no barrier after the prologue, every wave streams its own 16 PB pixels with the NEXT tile's operands in flight (a residual
//     chunk is requested into the registers its predecessor was just read from: a whole tile of look-ahead per wave);
//   * every vector-memory operation is a compiler-visible builtin (loads AND stores), so hipcc's own in-order vmcnt arithmetic is exact.
// Every output element is accumulated over K in one fixed order by one wave: bit-identical run to run, independent of the grid; and K
// is walked in the order the separate kernels walk it.
#include "common.h"

#ifndef PAIR_XAUX
#define PAIR_XAUX 0     // cache policy of the activation loads: bits 0-1 x (the 3x3 layer's output: this launch is its only reader), bits 2-3 x2 (the
                        // block input of the PROJ form: its last reader); buffer aux bits: 1 glc, 2 slc.  Measured in the network (2, 8, 10 against 0, two
                        // rounds): the pairs + 2..6 us, nothing gained behind them -- left at 0
#endif
namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bfround(float f) { return __uint_as_float(hsefr_bf16_bits(f) << 16); }

struct PairParams {
    const void* x;        // [M][K1] bf16
    const void* w1;       // [N1][K1] bf16
    const float* scale1;
    const float* shift1;
    const void* res;      // [M][N1] bf16 (null with PROJ)
    const void* x2;       // PROJ: [M][K2] bf16, the block input at the same pixels
    const void* wp;       // PROJ: [N1][K2] bf16
    const float* scale_p;
    const float* shift_p;
    void* y1;             // [M][N1] bf16
    const void* w2;       // [N2][N1] bf16
    const float* scale2;
    const float* shift2;
    void* y2;             // [M][N2] bf16
    unsigned M;
    unsigned ntiles;      // ceil(M / (16 PB))
    int reverse;
    float act1_lo, act1_hi, act2_lo, act2_hi;
    int ablate;           // development builds, timing only (results WRONG): 1 = no residual loads, 2 = no y1 stores, 4 = no y2 stores, 8 = no second product
    int nt;               // cache hints: 1 = y1 stores non-temporal, 2 = residual loads non-temporal, 4 = y2 stores non-temporal
    int y1_stride;        // 1: y1 [M][N1].  2: y1 is stored at the pixels with even row AND column only, as a compact [n][(H+1)/2][(W+1)/2][N1] map
                          // (the second product still reads every pixel, from registers): the tensor's only other reader takes every second pixel of
                          // it (the residual of a stage's last block, lowering.compact_pair_outputs) -- three quarters of the pair's largest store gone
    unsigned W, HW, OW2, OHW2;      // y1_stride 2: the map (W, H W), the compact map's width and pixels
    hsefr_udiv d_hw, d_w;           // ... exact division by H W and by W (common.h)
    long long y1_bytes;
};

// LDS row R of a weight image <-> output channel (conv1x1_w4_bf16.hip's permutation, per 32 channels): rows 16 b + i of a pair of
// 16-row blocks hold channel 8 (i >> 2) + 4 b + (i & 3), so that accumulator element e of block b in lane (l16, lq) is channel
// 8 lq + 4 b + e
__device__ __forceinline__ int perm_channel(int R) {
    const int i = R & 15, b = (R >> 4) & 1;
    return (R & ~31) + 8 * (i >> 2) + 4 * b + (i & 3);
}

template <int K1, int K2, int N1, int N2, int PB, int WAVES, bool PROJ>
__global__ __launch_bounds__(WAVES * 64) void conv1x1_pair_bf16_kernel(PairParams p) {
    static_assert(K1 % 64 == 0 && K2 % 64 == 0 && N1 % 64 == 0 && N2 % 32 == 0, "whole 64-channel K steps");
    constexpr int NT = WAVES * 64;
    constexpr int W1_OFF = 0, W1_BYTES = N1 * K1 * 2;
    constexpr int WP_OFF = W1_OFF + W1_BYTES, WP_BYTES = PROJ ? N1 * K2 * 2 : 0;
    constexpr int W2_OFF = WP_OFF + WP_BYTES, W2_BYTES = N2 * N1 * 2;
    constexpr int C_OFF = W2_OFF + W2_BYTES;            // constants: scale1 | shift1 | (scale_p | shift_p) | scale2 | shift2
    constexpr int C_BYTES = (2 * N1 + (PROJ ? 2 * N1 : 0) + 2 * N2) * 4;
    static_assert(C_OFF + C_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[C_OFF + C_BYTES];
    constexpr int H1 = K1 / 32, HP = K2 / 32, CH = N1 / 32, MB2 = N2 / 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lq = lane >> 4;

    // ---- prologue: the weight images and the constants, once per workgroup ----
    auto fill = [&](int off, const void* w, int rows, int k) __attribute__((always_inline)) {
        const int cpr = k / 8;                                     // 16-byte chunks per weight row
        for (int idx = tid; idx < rows * cpr; idx += NT) {
            const int R = idx / cpr, cc = idx - R * cpr;
            const int s = cc >> 3, c = cc & 7;
            const uint4 v = *(const uint4*)((const char*)w + ((size_t)perm_channel(R) * k + cc * 8) * 2);
            *(uint4*)(smem + off + s * rows * 128 + R * 128 + 16 * (c ^ (R & 6))) = v;
        }
    };
    fill(W1_OFF, p.w1, N1, K1);
    if (PROJ) fill(WP_OFF, p.wp, N1, K2);
    fill(W2_OFF, p.w2, N2, N1);
    {
        float* cst = (float*)(smem + C_OFF);
        for (int i = tid; i < N1; i += NT) {
            cst[i] = p.scale1[i];
            cst[N1 + i] = p.shift1[i];
            if (PROJ) {
                cst[2 * N1 + i] = p.scale_p[i];
                cst[3 * N1 + i] = p.shift_p[i];
            }
        }
        for (int i = tid; i < N2; i += NT) {
            cst[(PROJ ? 4 : 2) * N1 + i] = p.scale2[i];
            cst[(PROJ ? 4 : 2) * N1 + N2 + i] = p.shift2[i];
        }
    }
    __syncthreads();
    const float* cst = (const float*)(smem + C_OFF);
    constexpr int C_SP = 2 * N1, C_S2 = (PROJ ? 4 : 2) * N1;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (long long)p.M * K1 * 2);
    const __amdgpu_buffer_rsrc_t rx2 = make_rsrc(PROJ ? p.x2 : nullptr, PROJ ? (long long)p.M * K2 * 2 : 0);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(PROJ ? nullptr : p.res, PROJ ? 0 : (long long)p.M * N1 * 2);
    const __amdgpu_buffer_rsrc_t ry1 = make_rsrc(p.y1, p.y1_bytes);
    const __amdgpu_buffer_rsrc_t ry2 = make_rsrc(p.y2, (long long)p.M * N2 * 2);

    // fragment address of a weight image: row (16 blk + l16), K half hh (32 channels), swizzled like the four-wave GEMM's stages
    const unsigned fr0 = (unsigned)(l16 * 128 + 16 * (lq ^ (l16 & 6)));
    auto wfrag = [&](int off, int rows, int blk, int hh) __attribute__((always_inline)) -> bf16x8 {
        return *(const bf16x8*)(smem + off + (hh >> 1) * rows * 128 + blk * 16 * 128 + (fr0 ^ ((hh & 1) ? 64u : 0u)));
    };

    const unsigned gw = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
#ifdef HSEFR_DEV
    const int abl = p.ablate;
#else
    constexpr int abl = 0;
#endif
    constexpr unsigned OOR = 0x80000000u;        // beyond every resource (tensors < 2 GiB): loads return zeros, move no bytes
    // the lane's pixel in pixel block 0 of wave tile t (tiles past the last: out of range)
    auto tile_pix = [&](unsigned t) __attribute__((always_inline)) -> unsigned {
        const unsigned tt = p.reverse ? p.ntiles - 1u - t : t;
        return tt * (16u * PB) + (unsigned)l16;
    };
    // ---- a tile's operands are requested ONE TILE AHEAD: the residual chunk j of the next tile goes into the registers chunk j of this tile
    // has just been read from (CH chunks = a whole tile in flight per wave), the activations into a second set.  What the first version
    // (residual two chunks ahead, activations at the top of their tile) measured at batch 128: residual loads + y1 stores + compute
    // ADDED UP (157 us; 41 compute only, 89 without the loads, 78 without the stores) -- vmcnt retires in order, so every wait for a
    // young load also waited for the stores just before it, and a tile began by draining the previous tile's stores.
    bf16x8 xf[PB][H1], xn[PB][H1];
    bf16x8 x2f[PROJ ? PB : 1][PROJ ? HP : 1], x2n[PROJ ? PB : 1][PROJ ? HP : 1];
    f32x4 rr[PROJ ? 1 : CH][PB];
    auto load_x = [&](bf16x8 (&dst)[PB][H1], bf16x8 (&dst2)[PROJ ? PB : 1][PROJ ? HP : 1], unsigned t) __attribute__((always_inline)) {
        const bool live = t < p.ntiles;
        const unsigned pix0 = tile_pix(t);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
            for (int hh = 0; hh < H1; ++hh)
                dst[pb][hh] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, live ? (pix0 + 16u * pb) * (K1 * 2u) + (unsigned)(64 * hh + 16 * lq) : OOR, 0, PAIR_XAUX & 3));
            if (PROJ) {
#pragma unroll
                for (int hh = 0; hh < HP; ++hh)
                    dst2[PROJ ? pb : 0][PROJ ? hh : 0] =
                        __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx2, live ? (pix0 + 16u * pb) * (K2 * 2u) + (unsigned)(64 * hh + 16 * lq) : OOR, 0, (PAIR_XAUX >> 2) & 3));
            }
        }
    };
    auto load_res = [&](int j, unsigned t) __attribute__((always_inline)) {
        const unsigned base = t < p.ntiles ? tile_pix(t) * (N1 * 2u) + (unsigned)(16 * lq + 64 * j) : OOR;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const unsigned off = base == OOR ? OOR : base + (unsigned)(16 * pb * N1 * 2);
            rr[PROJ ? 0 : j][pb] = (p.nt & 2) ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, off, 0, 2)) : bload16(rres, off, 0);
        }
    };
    auto store16 = [&](f32x4 v, const __amdgpu_buffer_rsrc_t& r, unsigned off, bool nt) __attribute__((always_inline)) {
        if (nt) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(hsefr_u32x4, v), r, off, 0, 2);
            hsefr_store_guard();
        } else {
            bstore16(v, r, off, 0);
        }
    };
    load_x(xf, x2f, gw);
    if (!PROJ && !(abl & 1)) {
#pragma unroll
        for (int j = 0; j < CH; ++j) load_res(j, gw);
    }
    for (unsigned t = gw; t < p.ntiles; t += nw) {
        const unsigned pix0 = tile_pix(t);
        unsigned y1off[PB];      // byte offset of the lane's pixel of block pb in y1 (+ 64 j per chunk); out of range where the pixel is not stored
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const unsigned pix = pix0 + 16u * pb;
            if (p.y1_stride == 2) {      // (uniform)
                const unsigned img = hsefr_udiv_do(pix, p.d_hw), rem = pix - img * p.HW;
                const unsigned oy = hsefr_udiv_do(rem, p.d_w), ox = rem - oy * p.W;
                const bool keep = pix < p.M && ((oy | ox) & 1u) == 0u;
                y1off[pb] = keep ? (img * p.OHW2 + (oy >> 1) * p.OW2 + (ox >> 1)) * (N1 * 2u) + (unsigned)(16 * lq) : OOR;
            } else {
                y1off[pb] = pix * (N1 * 2u) + (unsigned)(16 * lq);
            }
        }
        f32x4 acc2[MB2][PB];
#pragma unroll
        for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) acc2[mb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (j == 1) load_x(xn, x2n, t + nw);                              // the next tile's activations
            f32x4 acc[2][PB];
            if (PROJ) {
                // the projected shortcut of these 32 channels: scaled, shifted, rounded to bf16 where its tensor used to be stored
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) acc[b][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int hh = 0; hh < HP; ++hh)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const bf16x8 a = wfrag(WP_OFF, N1, 2 * j + b, hh);
#pragma unroll
                        for (int pb = 0; pb < PB; ++pb)
                            acc[b][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, x2f[PROJ ? pb : 0][PROJ ? hh : 0], acc[b][pb], 0, 0, 0);
                    }
                f32x4 sc[2], sh[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    sc[b] = *(const f32x4*)(cst + C_SP + 32 * j + 8 * lq + 4 * b);
                    sh[b] = *(const f32x4*)(cst + C_SP + N1 + 32 * j + 8 * lq + 4 * b);
                }
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) {
                    float v[8];
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * b + e] = fmaf(acc[b][pb][e], sc[b][e], sh[b][e]);
#pragma unroll
                    for (int d = 0; d < 4; ++d) rr[0][pb][d] = __uint_as_float(hsefr_pack_bf16x2(v[2 * d], v[2 * d + 1]));
                }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) acc[b][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hh = 0; hh < H1; ++hh)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const bf16x8 a = wfrag(W1_OFF, N1, 2 * j + b, hh);
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) acc[b][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xf[pb][hh], acc[b][pb], 0, 0, 0);
                }
            // ---- epilogue of the chunk: Y1's 32 channels, stored AND kept (rounded) as the second product's B operand ----
            f32x4 sc[2], sh[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                sc[b] = *(const f32x4*)(cst + 32 * j + 8 * lq + 4 * b);
                sh[b] = *(const f32x4*)(cst + N1 + 32 * j + 8 * lq + 4 * b);
            }
            bf16x8 yf[PB];
            f32x4 o[PB];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                float v[8];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * b + e] = fmaf(acc[b][pb][e], sc[b][e], sh[b][e]);
                const f32x4 r = rr[PROJ ? 0 : j][pb];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned rw = __float_as_uint(r[d]);
                    v[2 * d] = bfround(v[2 * d]) + __uint_as_float(rw << 16);
                    v[2 * d + 1] = bfround(v[2 * d + 1]) + __uint_as_float(rw & 0xFFFF0000u);
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float f0 = fminf(fmaxf(v[2 * d], p.act1_lo), p.act1_hi), f1 = fminf(fmaxf(v[2 * d + 1], p.act1_lo), p.act1_hi);
                    o[pb][d] = __uint_as_float(hsefr_pack_bf16x2(f0, f1));
                }
                yf[pb] = __builtin_bit_cast(bf16x8, o[pb]);
            }
            if (!PROJ && !(abl & 1)) load_res(j, t + nw);                     // this chunk of the NEXT tile, into the registers just read
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
                if (!(abl & 2)) store16(o[pb], ry1, y1off[pb] + (unsigned)(64 * j), p.nt & 1);
            // ---- the chunk's contribution to every channel of Y2 ----
            if (!(abl & 8))
#pragma unroll
            for (int mb = 0; mb < MB2; ++mb) {
                const bf16x8 a = wfrag(W2_OFF, N2, mb, j);
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) acc2[mb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, yf[pb], acc2[mb][pb], 0, 0, 0);
            }
        }
        // ---- Y2's epilogue ----
        const unsigned y2lane = pix0 * (N2 * 2u) + (unsigned)(16 * lq);
#pragma unroll
        for (int m2 = 0; m2 < MB2 / 2; ++m2) {
            f32x4 sc[2], sh[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                sc[b] = *(const f32x4*)(cst + C_S2 + 32 * m2 + 8 * lq + 4 * b);
                sh[b] = *(const f32x4*)(cst + C_S2 + N2 + 32 * m2 + 8 * lq + 4 * b);
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                float v[8];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * b + e] = fmaf(acc2[2 * m2 + b][pb][e], sc[b][e], sh[b][e]);
                f32x4 o;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float f0 = fminf(fmaxf(v[2 * d], p.act2_lo), p.act2_hi), f1 = fminf(fmaxf(v[2 * d + 1], p.act2_lo), p.act2_hi);
                    o[d] = __uint_as_float(hsefr_pack_bf16x2(f0, f1));
                }
                if (!(abl & 4)) store16(o, ry2, y2lane + (unsigned)(16 * pb * N2 * 2 + 64 * m2), p.nt & 4);
            }
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
            for (int hh = 0; hh < H1; ++hh) xf[pb][hh] = xn[pb][hh];
            if (PROJ) {
#pragma unroll
                for (int hh = 0; hh < HP; ++hh) x2f[PROJ ? pb : 0][PROJ ? hh : 0] = x2n[PROJ ? pb : 0][PROJ ? hh : 0];
            }
        }
    }
}

HSEFR_KNOB(g_pair_off, 0);     // dev builds: 1 = the engine never pairs (A/B timing against the two launches)
HSEFR_KNOB(g_pair_ablate, 0);  // dev builds: PairParams::ablate
HSEFR_KNOB(g_pair_nt, 1);      // PairParams::nt.  1 = y1 with the non-temporal hint: 410 MB of y1 + residual stream through a pair while the NEXT launch
                               // (a 3x3) reads only the 51 MB of y2 -- measured in the network at batch 128, same box: the 3x3 behind a pair 47.3 -> 40.1 us,
                               // the stage's last increase layer 90 -> 78, the pairs themselves +7 / +8 us; ResNet-50 1.745 -> 1.730 ms (hints on the
                               // residual loads or on y2: slower)

template <int K1, int K2, int N1, int N2, int PB, int WAVES, bool PROJ>
int launch_pair(PairParams& p, hipStream_t s) {
    p.ntiles = (p.M + 16u * PB - 1u) / (16u * PB);
    const unsigned need = (p.ntiles + WAVES - 1) / WAVES;
    const unsigned g = need < 256u ? need : 256u;
    HSEFR_LAUNCH((conv1x1_pair_bf16_kernel<K1, K2, N1, N2, PB, WAVES, PROJ>), dim3(g), dim3(WAVES * 64), 0, s, p);
    return launch_status("conv1x1_pair_bf16");
}

}  // namespace

#ifdef HSEFR_DEV
void set_pair_off(int v) { g_pair_off = v; }
void set_pair_ablate(int v) { g_pair_ablate = v; }
void set_pair_nt(int v) { g_pair_nt = v; }
#endif

// c -> cout1 (+ residual, or + the projection of x2 [.., c2]) -> cout2, all at the same pixels
bool conv1x1_pair_bf16_shape_supported(int c, int cout1, int cout2, int c2) {
    return c == 64 && cout1 == 256 && cout2 == 64 && (c2 == 0 || c2 == 64);
}
bool conv1x1_pair_bf16_supported(long long pixels, int c, int cout1, int cout2, int c2) {
    if (g_pair_off == 1) return false;
    return pixels > 0 && pixels * (long long)cout1 * 2 < (1ll << 31) && conv1x1_pair_bf16_shape_supported(c, cout1, cout2, c2);
}

int launch_conv1x1_pair_bf16(const void* x, const void* w1, const float* scale1, const float* shift1, const void* res, const void* x2,
                             const void* wp, const float* scale_p, const float* shift_p, void* y1, const void* w2, const float* scale2,
                             const float* shift2, void* y2, long long pixels, int c, int cout1, int cout2, int c2, int act1, int act2,
                             hipStream_t s, int y1_stride, int h, int w) {
    HSEFR_REQUIRE(conv1x1_pair_bf16_supported(pixels, c, cout1, cout2, c2), HSEFR_ERR_UNSUPPORTED,
                  "conv1x1_pair_bf16: %d -> %d -> %d (projection from %d) over %lld pixels not covered", c, cout1, cout2, c2, pixels);
    HSEFR_REQUIRE((c2 > 0) != (res != nullptr), HSEFR_ERR_INVALID, "conv1x1_pair_bf16: exactly one of residual / projected shortcut");
    for (int a : {act1, act2})
        HSEFR_REQUIRE(a == HSEFR_ACT_NONE || a == HSEFR_ACT_RELU || a == HSEFR_ACT_RELU6, HSEFR_ERR_UNSUPPORTED, "conv1x1_pair_bf16: act %d", a);
    PairParams p;
    p.x = x; p.w1 = w1; p.scale1 = scale1; p.shift1 = shift1; p.res = res; p.x2 = x2; p.wp = wp; p.scale_p = scale_p; p.shift_p = shift_p;
    p.y1 = y1; p.w2 = w2; p.scale2 = scale2; p.shift2 = shift2; p.y2 = y2;
    p.M = (unsigned)pixels;
    p.y1_stride = y1_stride; p.y1_bytes = pixels * cout1 * 2;
    p.W = p.HW = p.OW2 = p.OHW2 = 0;
    p.d_hw = p.d_w = hsefr_udiv{0u, 0u};
    if (y1_stride != 1) {
        HSEFR_REQUIRE(y1_stride == 2 && h > 1 && w > 1 && pixels % ((long long)h * w) == 0, HSEFR_ERR_INVALID,
                      "conv1x1_pair_bf16: y1 at stride %d of a %dx%d map over %lld pixels", y1_stride, h, w, pixels);
        p.W = (unsigned)w; p.HW = (unsigned)(h * w);
        p.d_hw = hsefr_udiv_make(p.HW); p.d_w = hsefr_udiv_make(p.W);
        p.OW2 = (unsigned)((w + 1) / 2); p.OHW2 = (unsigned)(((h + 1) / 2) * ((w + 1) / 2));
        p.y1_bytes = pixels / ((long long)h * w) * p.OHW2 * cout1 * 2;
    }
    p.reverse = sweep_reverse();
    p.ablate = g_pair_ablate;
    p.nt = g_pair_nt;
    p.act1_lo = act1 == HSEFR_ACT_NONE ? -INFINITY : 0.f;
    p.act1_hi = act1 == HSEFR_ACT_RELU6 ? 6.f : INFINITY;
    p.act2_lo = act2 == HSEFR_ACT_NONE ? -INFINITY : 0.f;
    p.act2_hi = act2 == HSEFR_ACT_RELU6 ? 6.f : INFINITY;
    if (c2 > 0) return launch_pair<64, 64, 256, 64, 2, 8, true>(p, s);
    return launch_pair<64, 64, 256, 64, 2, 8, false>(p, s);
}

}  // namespace hsefr
