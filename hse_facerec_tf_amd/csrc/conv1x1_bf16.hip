// 1x1 stride-1 bf16 convolution (+ BatchNorm scale/shift, optional residual add, ReLU) as a PERSISTENT bf16-MFMA GEMM,
// NHWC bf16 in / bf16 out, gfx950: the "reduce" and "increase" layers of ResNet-50's bottlenecks (resnet50_ft,
// vgg2_resnet.pb at facerec_test.py:213), 54 % of that model's time.
//
//   Y[p, n] = act( bf16( scale[n] * sum_c X[p, c] * Wt[n, c] + shift[n] ) (+ R[p, n]) )        (same rounding points as
//                                                                                               conv_bf16.hip)
// These layers have one to sixteen K-tiles and as many residual/output bytes as input bytes: memory-bound.  The general
// implicit-GEMM kernel (conv_bf16.hip) runs one tile per workgroup and exposes every load latency; this one is the
// skeleton of the split-f16 pointwise GEMM (pwconv_f16s.hip) without the split:
//   * persistent workgroups over a flat (tile, K-tile) step sequence, raw-buffer loads TWO steps ahead (two register sets,
//     one VGPR offset for the whole kernel, rows beyond P dropped / zero-filled by the hardware);
//   * 128-B LDS rows (64 bf16 = one K-tile) with the (row >> 1) & 7 chunk swizzle, double-buffered stages;
//   * v_mfma_f32_32x32x16_bf16 with the operands swapped (weights first), so a lane ends up with 4 consecutive channels;
//   * the tile's RESIDUAL chunks are requested at the tile's first step (registers) and are long there when the epilogue
//     needs them; the epilogue transposes through a wave-private LDS scratch (the stage just consumed) and leaves as
//     16-byte stores of whole 128-B row segments.
//
// PROJ (round 5): the first block of a ResNet stage adds a PROJECTED shortcut -- a second 1x1 convolution (stride 1 or 2, its own
// BatchNorm) of the block's input.  As two launches the projection's [P, Cout] tensor is written and read back (205 MB of the
// 719 MB the pair moves in stage 2); here the tile's K loop simply starts with the projection's K-tiles (activation rows GATHERED
// from the strided input pixels, the projection's weight rows), its result is rounded to bf16 where the tensor used to be
// stored -- the same rounding points: oracle/resnet50.py -- and parked in 32 registers per lane, and the main product's epilogue
// adds it:   Y = act( bf16( s W x + b ) + bf16( s2 W2 x2[stride] + b2 ) ).
#include <type_traits>

#include "common.h"

namespace hsefr {

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

__device__ __forceinline__ u16 f2bf(float f) { return (u16)hsefr_bf16_bits(f); }      // round-to-nearest-even (common.h)
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ int swzb(int row, int chunk) { return row * 128 + 16 * (chunk ^ ((row >> 1) & 7)); }

#ifndef C11_RESAUX
#define C11_RESAUX 2     // cache policy of the residual loads (buffer aux bits: 1 glc, 2 slc): this layer is the residual's LAST reader -- streamed (slc), it leaves the caches to the output the next two layers read: ResNet-50 1.650 -> 1.635 ms per 128, same box
#endif
__device__ __forceinline__ bf16x8 bload8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore8(bf16x8 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(hsefr_u32x4, v), r, voff, soff, 0);
    hsefr_store_guard();
}

#ifdef HSEFR_CD_STAMPS
// Diagnostic build only: per-wave s_memtime sums -- [0] loads issued + fragment reads + MFMAs, [1] LDS stage writes (the wait for the
// global loads of two steps ago sits here), [2] step barrier, [3] epilogue, [4] the barrier behind it; [6] lifetime, [7] steps
__device__ unsigned long long g_c11_stamps[512 * 4 * 8];
#define C11_STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); st[i] += _t - tprev; tprev = _t; } while (0)
#define C11_STAMP_DECL unsigned long long st[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tprev = __builtin_amdgcn_s_memtime(); const unsigned long long tstart = tprev
#define C11_STAMP_FLUSH do { if (lane == 0 && blockIdx.x < 512) { unsigned long long* o = g_c11_stamps + (blockIdx.x * 4 + wave) * 8; \
    for (int i_ = 0; i_ < 6; ++i_) o[i_] = st[i_]; o[6] = __builtin_amdgcn_s_memtime() - tstart; o[7] = nsteps; } } while (0)
#else
#define C11_STAMP(i) do { } while (0)
#define C11_STAMP_DECL do { } while (0)
#define C11_STAMP_FLUSH do { } while (0)
#endif

struct ProjParams {        // the projected shortcut of PROJ kernels
    const u16* x2;         // [N, H2, W2, K2] bf16: the block's input
    const u16* wt2;        // [Cout][K2]
    const float* scale2;   // [Cout]
    const float* shift2;   // [Cout]
    int K2, stride, H2, W2, OH, OW;
    long long x2_bytes;
    int rs_stride;         // (plain RES kernels; round 6) > 0: the residual is read from a LARGER map -- row p = (img, oy, ox) of the output adds
    hsefr_udiv d_ohow, d_ow; // res[img, oy * rs_stride, ox * rs_stride, :] of a [N, H2, W2, Cout] tensor of x2_bytes bytes (H2, W2, OH, OW above;
                           // d_*: exact division by OH * OW and by OW, common.h).  The last block of a ResNet stage computed only at the
                           // pixels the next stage's stride-2 layers read (lowering.subsample_stage_tails)
    int adv;               // (both kernel forms; round 6) rows a tile ADVANCES by, <= BM: the tile computes BM rows but owns -- loads, adds the
                           // residual of, stores -- only the first `adv` (every resource ends with the tile's last own row: what lies beyond
                           // reads as zeros and moves no bytes).  Chosen by the launcher so that the tiles fill whole rounds of the grid's
                           // slots: 14 x 14 x 1024 at batch 128 is 1568 tiles of 128 rows = 3.06 rounds that cost 4; as 2048 tiles of 98 rows
                           // it is 4.0 rounds of 0.77 of the memory work each (these layers are bound by what a CU's memory path moves)
    int b_resident;        // (both kernel forms) 1: a tile's K loop is at most two steps AND every tile of a workgroup has the same channel
                           // origin -- each LDS stage then always holds the SAME weight tile: it is loaded with the first two steps and
                           // never again (a quarter to a half of the kernel's load instructions on the K <= 128 layers, which are bound
                           // by what they issue: DESIGN.md lesson 54)
};

template <int BM, int BN, int OCC, bool RES, int ACT, bool PROJ = false>
__global__ __launch_bounds__(256, OCC) void conv1x1_bf16_kernel(const u16* __restrict__ x, const u16* __restrict__ wt,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const u16* __restrict__ res, u16* __restrict__ y, long long P,
                                                                int K, int Cout, unsigned tiles_n, unsigned total_tiles,
                                                                int reverse, ProjParams pj) {
    static_assert(!(PROJ && RES), "the projected shortcut takes the residual's place");
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int AP = BM / 32, BP = BN / 32;
    static_assert(WN == 64 || WN == 32, "wave tile width");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][(BM + BN) * 128];
    __shared__ __attribute__((aligned(16))) float Et[2][PROJ ? 4 : 2][BN];   // [tile parity][scale | shift (| scale2 | shift2)][n]
    auto As = [&](int st) { return &smem[st][0]; };
    auto Bs = [&](int st) { return &smem[st][BM * 128]; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int srow = tid >> 3, sch = tid & 7;
    const int KT2 = PROJ ? pj.K2 / 64 : 0;              // a tile's K loop: the projection's K-tiles first, then the main product's
    const int KT = K / 64 + KT2;
    if (blockIdx.x >= total_tiles) return;
    const unsigned ntile = (total_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const unsigned nsteps = ntile * KT;
    const unsigned rowbytes = (unsigned)K * 2u;
    const unsigned voff = (unsigned)srow * rowbytes + 16u * sch;
    const unsigned rowbytes2 = PROJ ? (unsigned)pj.K2 * 2u : 0u;
    const unsigned voff2b = (unsigned)srow * rowbytes2 + 16u * sch;          // projection weight rows
    unsigned avoff2[PROJ ? AP : 1];                                          // gathered projection input rows of the prefetch cursor's tile

    auto tile_origin = [&](unsigned i, long long& mm0, int& nn0) {
        const unsigned lt = xcd_remap_dir(blockIdx.x + (i < ntile ? i : ntile - 1) * gridDim.x, total_tiles, reverse);
        mm0 = (long long)(lt / tiles_n) * pj.adv;
        nn0 = (lt % tiles_n) * BN;
    };
    auto own_end = [&](long long mm0) { return mm0 + pj.adv < P ? mm0 + pj.adv : P; };      // one past the tile's last own row
    __amdgpu_buffer_rsrc_t ra_rsrc, rb_rsrc, rb2_rsrc;
    const __amdgpu_buffer_rsrc_t ra2_rsrc = make_rsrc(PROJ ? pj.x2 : nullptr, PROJ ? pj.x2_bytes : 0);
    unsigned pf_i = 0;
    int pf_kt = 0;
    auto setup_rsrc = [&](unsigned i) {
        long long mm0;
        int nn0;
        tile_origin(i, mm0, nn0);
        ra_rsrc = make_rsrc(x + mm0 * K, (own_end(mm0) - mm0) * (long long)rowbytes);
        rb_rsrc = make_rsrc(wt + (long long)nn0 * K, (long long)(Cout - nn0) * rowbytes);
        if (PROJ) {
            rb2_rsrc = make_rsrc(pj.wt2 + (long long)nn0 * pj.K2, (long long)(Cout - nn0) * rowbytes2);
            const unsigned ohow = (unsigned)(pj.OH * pj.OW);
#pragma unroll
            for (int p = 0; p < AP; ++p) {
                const long long m = mm0 + srow + 32 * p;                 // output pixel -> the input pixel it is projected from
                const unsigned mu = (unsigned)(m < P ? m : 0);
                const unsigned n = mu / ohow, rem = mu - n * ohow;
                const unsigned oh = rem / (unsigned)pj.OW, ow = rem - oh * (unsigned)pj.OW;
                const unsigned pix = (n * (unsigned)pj.H2 + oh * (unsigned)pj.stride) * (unsigned)pj.W2 + ow * (unsigned)pj.stride;
                avoff2[PROJ ? p : 0] = m < own_end(mm0) ? pix * rowbytes2 + 16u * sch : 0x80000000u;
            }
        }
    };
    bf16x8 ra[2][AP], rb[2][BP];
    const bool b_res = pj.b_resident != 0;
    auto gload = [&](auto SET, bool with_b) {
        constexpr int S = decltype(SET)::value;
        if (PROJ && pf_kt < KT2) {                                       // (wave-uniform)
            const unsigned so = (unsigned)pf_kt * 128u;
#pragma unroll
            for (int p = 0; p < AP; ++p) ra[S][p] = bload8(ra2_rsrc, avoff2[PROJ ? p : 0], so);
            if (with_b) {
#pragma unroll
                for (int p = 0; p < BP; ++p) rb[S][p] = bload8(rb2_rsrc, voff2b, so + (unsigned)(32 * p) * rowbytes2);
            }
            return;
        }
        const unsigned so = (unsigned)(pf_kt - KT2) * 128u;
#pragma unroll
        for (int p = 0; p < AP; ++p) ra[S][p] = bload8(ra_rsrc, voff, so + (unsigned)(32 * p) * rowbytes);
        if (with_b) {
#pragma unroll
            for (int p = 0; p < BP; ++p) rb[S][p] = bload8(rb_rsrc, voff, so + (unsigned)(32 * p) * rowbytes);
        }
    };
    auto advance_prefetch = [&]() {
        if (++pf_kt == KT) {
            pf_kt = 0;
            setup_rsrc(++pf_i);
        }
    };
    auto swrite = [&](auto SET, int buf, bool with_b) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int p = 0; p < AP; ++p) *(bf16x8*)(As(buf) + swzb(srow + 32 * p, sch)) = ra[S][p];
        if (with_b) {
#pragma unroll
            for (int p = 0; p < BP; ++p) *(bf16x8*)(Bs(buf) + swzb(srow + 32 * p, sch)) = rb[S][p];
        }
    };

    f32x16 acc[NI][MI];
    auto zero_acc = [&]() {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
    };
    zero_acc();
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    long long m0;
    int n0;
    unsigned ci = 0;
    int ckt = 0;
    tile_origin(0, m0, n0);
    setup_rsrc(0);
    gload(S0(), true);
    advance_prefetch();
    gload(S1(), true);
    advance_prefetch();
    swrite(S0(), 0, true);
    __syncthreads();
    unsigned gstep = 0;            // steps done: step g loads step g + 2 and parks step g + 1 -- with resident weights only steps 0, 1 carry a weight tile
    const int xrow = wm * WM + li, wrow = wn * WN + li;
    // epilogue geometry (after the transpose): a wave's tile row is WN bf16 = WN*2 bytes = CPR chunks of 16 B;
    // lane -> (row erow of a 32-row block, chunk ech); 64 / CPR rows per store instruction
    constexpr int CPR = WN / 8;
    constexpr int RPI = 64 / CPR;
    const int erow = lane / CPR, ech = lane % CPR;
    const unsigned yvoff = ((unsigned)(wm * WM + erow) * (unsigned)Cout + (unsigned)(wn * WN + 8 * ech)) * 2u;
    bf16x8 rres[RES ? MI * (32 / RPI) : 1];
    unsigned pk[PROJ ? NI : 1][PROJ ? MI : 1][PROJ ? 8 : 1];      // the projected shortcut of the tile, bf16 pairs in the accumulators' layout

    C11_STAMP_DECL;
    auto step = [&](auto PAR) {
        constexpr int PB = decltype(PAR)::value;
        const bool first = ckt == 0;
        f32x4 ec;
        const bool fill = first && tid < (PROJ ? BN : BN / 2);
        const int etab = tid / (BN / 4), ej = tid % (BN / 4);      // table 0..3 = scale | shift | scale2 | shift2, float4 index
        if (fill) {
            const float* src = etab == 0 ? scale : etab == 1 ? shift : etab == 2 ? pj.scale2 : pj.shift2;
            ec = *(const f32x4*)(src + n0 + 4 * ej);
        }
        if (RES && first && pj.rs_stride > 0) {      // the residual's pixels are every rs_stride-th of a larger map: one address per row (uniform branch)
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(res, pj.x2_bytes);
            const unsigned ohow = (unsigned)(pj.OH * pj.OW), pend = (unsigned)own_end(m0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int i = 0; i < 32 / RPI; ++i) {
                    const unsigned pr = (unsigned)m0 + (unsigned)(wm * WM + mi * 32 + erow + RPI * i);
                    const unsigned img = hsefr_udiv_do(pr, pj.d_ohow), rem = pr - img * ohow;
                    const unsigned oy = hsefr_udiv_do(rem, pj.d_ow), ox = rem - oy * (unsigned)pj.OW;
                    const unsigned pix = (img * (unsigned)pj.H2 + oy * (unsigned)pj.rs_stride) * (unsigned)pj.W2 + ox * (unsigned)pj.rs_stride;
                    const unsigned off = (pix * (unsigned)Cout + (unsigned)(n0 + wn * WN + 8 * ech)) * 2u;
                    rres[RES ? mi * (32 / RPI) + i : 0] =
                        __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rr, pr < pend ? off : 0x80000000u, 0, C11_RESAUX));
                }
        } else if (RES && first) {      // this tile's residual chunks, in the layout the epilogue stores in (uniform branch)
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(res + m0 * Cout + n0, ((own_end(m0) - m0) * Cout - n0) * 2ll);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int i = 0; i < 32 / RPI; ++i)
                    rres[RES ? mi * (32 / RPI) + i : 0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rr, yvoff, (unsigned)(mi * 32 + RPI * i) * (unsigned)Cout * 2u, C11_RESAUX));
        }
        gload(PAR, !b_res);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x8 xa[MI], wb[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xa[mi] = *(const bf16x8*)(As(PB) + swzb(xrow + mi * 32, 2 * q + lh));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) wb[ni] = *(const bf16x8*)(Bs(PB) + swzb(wrow + ni * 32, 2 * q + lh));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
        }
        C11_STAMP(0);
        swrite(std::integral_constant<int, 1 - PB>(), 1 - PB, !b_res || gstep == 0);
        ++gstep;
        if (fill) *(f32x4*)(&Et[ci & 1][etab][4 * ej]) = ec;
        C11_STAMP(1);
        __syncthreads();
        C11_STAMP(2);
        advance_prefetch();
        ++ckt;
        if (PROJ && ckt == KT2) {
            // the projection is complete: scale2 / shift2, rounded to bf16 (where its tensor used to be stored), parked; the
            // accumulators start again for the main product
            const float* et = &Et[ci & 1][0][0];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch = ni * 32 + 8 * g + 4 * lh;
                        const f32x4 sc = *(const f32x4*)(et + 2 * BN + wn * WN + ch), sh = *(const f32x4*)(et + 3 * BN + wn * WN + ch);
                        pk[PROJ ? ni : 0][PROJ ? mi : 0][PROJ ? 2 * g : 0] =
                            hsefr_pack_bf16x2(fmaf(acc[ni][mi][4 * g + 0], sc[0], sh[0]), fmaf(acc[ni][mi][4 * g + 1], sc[1], sh[1]));
                        pk[PROJ ? ni : 0][PROJ ? mi : 0][PROJ ? 2 * g + 1 : 0] =
                            hsefr_pack_bf16x2(fmaf(acc[ni][mi][4 * g + 2], sc[2], sh[2]), fmaf(acc[ni][mi][4 * g + 3], sc[3], sh[3]));
                    }
            zero_acc();
        }
        if (ckt == KT) {
            const float* et = &Et[ci & 1][0][0];
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + m0 * Cout + n0, ((own_end(m0) - m0) * Cout - n0) * 2ll);
            unsigned char* scr = &smem[PB][wave * 4096];       // stage PB: every wave is past its last read (barrier above)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                // D[row = channel][col = pixel]: lane -> pixel li, registers 4g..4g+3 -> channels ni*32 + 8g + 4*lh + 0..3
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch = ni * 32 + 8 * g + 4 * lh;          // within the wave's WN columns
                        const f32x4 sc = *(const f32x4*)(et + wn * WN + ch), sh = *(const f32x4*)(et + BN + wn * WN + ch);
                        ushort4 o;
                        o.x = f2bf(fmaf(acc[ni][mi][4 * g + 0], sc[0], sh[0]));
                        o.y = f2bf(fmaf(acc[ni][mi][4 * g + 1], sc[1], sh[1]));
                        o.z = f2bf(fmaf(acc[ni][mi][4 * g + 2], sc[2], sh[2]));
                        o.w = f2bf(fmaf(acc[ni][mi][4 * g + 3], sc[3], sh[3]));
                        if (PROJ) {      // + the parked shortcut, activation, second rounding (the residual path of the plain kernel)
                            const unsigned q0 = pk[PROJ ? ni : 0][PROJ ? mi : 0][PROJ ? 2 * g : 0], q1 = pk[PROJ ? ni : 0][PROJ ? mi : 0][PROJ ? 2 * g + 1 : 0];
                            o.x = f2bf(apply_act<ACT>(bf2f(o.x) + __uint_as_float(q0 << 16)));
                            o.y = f2bf(apply_act<ACT>(bf2f(o.y) + __uint_as_float(q0 & 0xFFFF0000u)));
                            o.z = f2bf(apply_act<ACT>(bf2f(o.z) + __uint_as_float(q1 << 16)));
                            o.w = f2bf(apply_act<ACT>(bf2f(o.w) + __uint_as_float(q1 & 0xFFFF0000u)));
                        }
                        // scratch row li (WN*2 bytes), 16-B chunk (ch >> 3) swizzled by the row, 8-B half (ch >> 2) & 1
                        *(ushort4*)(scr + li * (WN * 2) + 16 * ((ch >> 3) ^ (li % CPR)) + 8 * ((ch >> 2) & 1)) = o;
                    }
#pragma unroll
                for (int i = 0; i < 32 / RPI; ++i) {
                    const int r = erow + RPI * i;
                    bf16x8 v = *(const bf16x8*)(scr + r * (WN * 2) + 16 * (ech ^ (r % CPR)));
                    if (!PROJ && (RES || ACT != HSEFR_ACT_NONE)) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float f = bf2f((u16)v[e]);
                            if (RES) f = f + bf2f((u16)rres[RES ? mi * (32 / RPI) + i : 0][e]);
                            f = apply_act<ACT>(f);
                            v[e] = (short)f2bf(f);
                        }
                    }
                    // rows beyond P fall outside the resource and are dropped by the hardware
                    bstore8(v, ry, yvoff, (unsigned)(mi * 32 + RPI * i) * (unsigned)Cout * 2u);
                }
            }
            C11_STAMP(3);
            __syncthreads();   // the scratch is the stage the next step refills
            C11_STAMP(4);
            zero_acc();
            ckt = 0;
            ++ci;
            tile_origin(ci, m0, n0);
        }
    };
    for (unsigned g = 0; g < nsteps; g += 2) {
        step(S0());
        if (g + 1 >= nsteps) break;
        step(S1());
    }
    C11_STAMP_FLUSH;
}

HSEFR_KNOB(g_c11_bres, 1);   // dev builds: 0 = reload the weight tile every step also where it could stay resident (A/B timing)
HSEFR_KNOB(g_c11_adv, 1);    // 1 = always BM: the product's tiling; dev builds: 0 = choose_adv's estimate, other = that many rows.
                             // MEASURED AND LOST (round 6, ResNet-50 batch 128 in the network, same box): the 14 x 14 x 1024 increase layers as
                             // 2048 tiles of 98 own rows (4.0 rounds of 512 slots) 35.5 us against 32.5 us as 1568 tiles of 128 (3.06 rounds);
                             // 112 and 120 rows: 32.3-33.7; the 28 x 28 x 512 layers 45-47 against 43-44.  The "3.06 rounds cost 4" reading of
                             // round 5 was wrong: persistent workgroups, two per CU, do not run in rounds -- the partial last round overlaps
                             // the one before it, and a shorter tile only adds tiles (weight tile, constants, barriers per tile)

// rows per tile (ProjParams::adv): the candidate in [BM / 2, BM] with the lowest estimated cost = rounds of the grid's slots x the
// share of a full tile's time a tile of that many own rows takes (FIXED of it does not shrink with the rows: the MFMAs run on all BM
// rows, the weight tile is loaded whole); ties go to the taller tile
template <int BM>
int choose_adv(long long P, unsigned tiles_n, long long slots) {
    if (g_c11_adv == 1) return BM;
    if (g_c11_adv > 1) return g_c11_adv < BM ? g_c11_adv : BM;
    constexpr double FIXED = 0.3;
    int best = BM;
    double best_cost = 1e300;
    for (int adv = BM; adv >= BM / 2; --adv) {
        const long long tiles = ((P + adv - 1) / adv) * tiles_n;
        if (tiles <= slots) return best_cost < 1e300 ? best : BM;          // a single (partial) round: the tallest tile that still is one
        const double cost = (double)((tiles + slots - 1) / slots) * (FIXED + (1.0 - FIXED) * adv / BM);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = adv; }
    }
    return best;
}

template <int BM, int BN, int OCC>
int launch_cfg(const u16* x, const u16* wt, const float* scale, const float* shift, const u16* res, u16* y, long long P, int K,
               int cout, int act, hipStream_t s, const ProjParams* sres) {
    const unsigned tiles_n = cout / BN;
    const long long slots = 256ll * OCC;
    const int adv = choose_adv<BM>(P, tiles_n, slots);
    const long long tiles_m = (P + adv - 1) / adv;
    const long long total = tiles_m * tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv1x1_bf16: too many tiles");
    const long long g = total < slots ? total : slots;
    dim3 grid((unsigned)g), block(256);
    const int rev = sweep_reverse();
    ProjParams nopj{};
    if (sres) nopj = *sres;      // (strided residual: rs_stride, H2, W2, OH, OW, m_*, x2_bytes)
    nopj.adv = adv;
    // resident weight tiles: K <= 128 (at most two steps per tile) and a grid whose stride keeps a workgroup on one channel origin
    // (and BM >= 128: the epilogue's 16 KB of wave-private scratch must fit the stage's ACTIVATION rows, or it lands on the weights)
    nopj.b_resident = (BM >= 128 && total > g && K <= 128 && g % 8 == 0 && (g / 8) % tiles_n == 0 && g_c11_bres) ? 1 : 0;
#define HSEFR_C11(R, A) HSEFR_LAUNCH((conv1x1_bf16_kernel<BM, BN, OCC, R, A>), grid, block, 0, s, x, wt, scale, shift, res, y, P, K, cout, tiles_n, (unsigned)total, rev, nopj)
    if (res) {
        if (act == HSEFR_ACT_RELU) HSEFR_C11(true, HSEFR_ACT_RELU);
        else if (act == HSEFR_ACT_RELU6) HSEFR_C11(true, HSEFR_ACT_RELU6);
        else HSEFR_C11(true, HSEFR_ACT_NONE);
    } else {
        if (act == HSEFR_ACT_RELU) HSEFR_C11(false, HSEFR_ACT_RELU);
        else if (act == HSEFR_ACT_RELU6) HSEFR_C11(false, HSEFR_ACT_RELU6);
        else HSEFR_C11(false, HSEFR_ACT_NONE);
    }
#undef HSEFR_C11
    return launch_status("conv1x1_bf16");
}

template <int BM, int BN, int OCC>
int launch_proj_cfg(const u16* x, const u16* wt, const float* scale, const float* shift, u16* y, long long P, int K, int cout, int act,
                    const ProjParams& pj, hipStream_t s) {
    const unsigned tiles_n = cout / BN;
    const long long slots = 256ll * OCC;
    const int adv = choose_adv<BM>(P, tiles_n, slots);
    const long long tiles_m = (P + adv - 1) / adv;
    const long long total = tiles_m * tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv1x1_proj_bf16: too many tiles");
    const long long g = total < slots ? total : slots;
    dim3 grid((unsigned)g), block(256);
    const int rev = sweep_reverse();
    ProjParams pjr = pj;
    pjr.adv = adv;
    pjr.b_resident = (BM >= 128 && total > g && K + pj.K2 <= 128 && g % 8 == 0 && (g / 8) % tiles_n == 0 && g_c11_bres) ? 1 : 0;
#define HSEFR_C11P(A) HSEFR_LAUNCH((conv1x1_bf16_kernel<BM, BN, OCC, false, A, true>), grid, block, 0, s, x, wt, scale, shift, nullptr, y, P, K, cout, tiles_n, (unsigned)total, rev, pjr)
    if (act == HSEFR_ACT_RELU) HSEFR_C11P(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_RELU6) HSEFR_C11P(HSEFR_ACT_RELU6);
    else HSEFR_C11P(HSEFR_ACT_NONE);
#undef HSEFR_C11P
    return launch_status("conv1x1_proj_bf16");
}

HSEFR_KNOB(g_c11_tile, 0);   // dev builds: 1 = 128 x 64 tiles (three workgroups per CU) everywhere, 2 = 128 x 128 everywhere
HSEFR_KNOB(g_c11, 1);   // dev builds: 0 = route 1x1 stride-1 layers through the general conv_bf16 kernel (A/B timing)

}  // namespace

#ifdef HSEFR_DEV
int read_c11_stamps(void* host_out, size_t bytes) {
#ifdef HSEFR_CD_STAMPS
    HSEFR_REQUIRE(bytes <= sizeof(unsigned long long) * 512 * 4 * 8, HSEFR_ERR_INVALID, "read_c11_stamps: too many bytes");
    HSEFR_HIP_CHECK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_c11_stamps), bytes));
    return HSEFR_OK;
#else
    (void)host_out; (void)bytes;
    set_error("read_c11_stamps: library built without -DHSEFR_CD_STAMPS");
    return HSEFR_ERR_UNSUPPORTED;
#endif
}
void set_c11(int v) { g_c11 = v; }
void set_c11_tile(int v) { g_c11_tile = v; }
void set_c11_bres(int v) { g_c11_bres = v; }
void set_c11_adv(int v) { g_c11_adv = v; }
#endif
bool conv1x1_bf16_enabled(bool has_res, int k, int cout) {
    switch (g_c11) {      // values > 1: bisection aids
        case 0: return false;
        case 2: return has_res;
        case 3: return !has_res;
        case 4: return k == 64;
        case 5: return k > 64 && cout <= 256;
        case 6: return k > 64 && cout > 256;
        default: return true;
    }
}

// x [P][K] bf16, wt [cout][K] bf16, y [P][cout] bf16; K % 64 == 0, cout % 64 == 0 (checked by the caller, launch_conv_bf16)
static int launch_c11(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y,
                      long long P, int K, int cout, int act, hipStream_t s, const ProjParams* sres) {
    const u16* xx = (const u16*)x;
    const u16* ww = (const u16*)wt;
    const u16* rr = (const u16*)res;
    u16* yy = (u16*)y;
    // 128 x 128 tiles when there are enough of them to fill the machine twice over, 128 x 64 otherwise
    const long long t128 = ((P + 127) / 128) * (cout / 128);
    // (dev builds: 64 x 128 tiles, three workgroups per CU.  Measured on the short-K increase layers: 3-7 % SLOWER than 128 x 128 -- its 16 KB
    // of epilogue scratch does not fit beside resident weights in a 64-row stage, and with the weights re-loaded every step it loses)
    if (cout % 128 == 0 && g_c11_tile == 3) return launch_cfg<64, 128, 3>(xx, ww, scale, shift, rr, yy, P, K, cout, act, s, sres);
    if (cout % 128 == 0 && (g_c11_tile == 2 || (t128 >= 768 && g_c11_tile != 1))) return launch_cfg<128, 128, 2>(xx, ww, scale, shift, rr, yy, P, K, cout, act, s, sres);
    return launch_cfg<128, 64, 3>(xx, ww, scale, shift, rr, yy, P, K, cout, act, s, sres);
}

int launch_conv1x1_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y,
                        long long P, int K, int cout, int act, hipStream_t s) {
    return launch_c11(x, wt, scale, shift, res, y, P, K, cout, act, s, nullptr);
}

// The same layer with the residual read from a LARGER map at every `stride`-th pixel (round 6):
//   y[(i, oy, ox), :] = act( bf16( scale * (x[(i, oy, ox), :] . wt) + shift ) + res[i, oy * stride, ox * stride, :] )
// x [n][oh][ow][K], res [n][h2][w2][cout]: the last block of a ResNet stage computed only at the pixels the next stage's stride-2 layers
// read (lowering.subsample_stage_tails) -- its shortcut is still the full-size map of the block before it.
int launch_conv1x1_sres_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y, int n, int oh,
                             int ow, int K, int cout, int stride, int h2, int w2, int act, hipStream_t s) {
    HSEFR_REQUIRE(K > 0 && K % 64 == 0 && cout > 0 && cout % 64 == 0, HSEFR_ERR_UNSUPPORTED, "conv1x1_sres_bf16: channels %d -> %d must be multiples of 64", K, cout);
    HSEFR_REQUIRE(res && n >= 0 && oh > 0 && ow > 0 && stride >= 1 && stride <= 3 && (oh - 1) * stride < h2 && (ow - 1) * stride < w2, HSEFR_ERR_INVALID,
                  "conv1x1_sres_bf16: a %dx%d output is not a stride-%d view of a %dx%d residual", oh, ow, stride, h2, w2);
    HSEFR_REQUIRE(act == HSEFR_ACT_NONE || act == HSEFR_ACT_RELU || act == HSEFR_ACT_RELU6, HSEFR_ERR_UNSUPPORTED, "conv1x1_sres_bf16: act %d", act);
    if (n == 0) return HSEFR_OK;
    const long long P = (long long)n * oh * ow;
    ProjParams g{};
    g.rs_stride = stride; g.H2 = h2; g.W2 = w2; g.OH = oh; g.OW = ow;
    g.x2_bytes = (long long)n * h2 * w2 * cout * 2;
    HSEFR_REQUIRE(P < (1ll << 31) && g.x2_bytes < (1ll << 32), HSEFR_ERR_UNSUPPORTED, "conv1x1_sres_bf16: tensors beyond the 32-bit offsets");
    HSEFR_REQUIRE(oh * ow > 1 && ow > 1, HSEFR_ERR_UNSUPPORTED, "conv1x1_sres_bf16: maps of one pixel / one column");
    g.d_ohow = hsefr_udiv_make((unsigned)(oh * ow));
    g.d_ow = hsefr_udiv_make((unsigned)ow);
    return launch_c11(x, wt, scale, shift, res, y, P, K, cout, act, s, &g);
}

// The increase layer of a stage's first block with its projected shortcut in the same launch (PROJ, see the header):
//   y[p, :] = act( bf16( scale * (x[p, :] . wt) + shift ) + bf16( scale2 * (x2[pixel(p) * stride, :] . wt2) + shift2 ) )
// x [P][K], wt [cout][K], x2 [n][h2][w2][k2], wt2 [cout][k2], y [P][cout]; P = n * oh * ow, output pixel (oh, ow) is projected from
// input pixel (oh * stride, ow * stride).
int launch_conv1x1_proj_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* x2, const void* wt2,
                             const float* scale2, const float* shift2, void* y, int n, int oh, int ow, int K, int cout, int k2, int stride,
                             int h2, int w2, int act, hipStream_t s) {
    HSEFR_REQUIRE(K > 0 && K % 64 == 0 && k2 > 0 && k2 % 64 == 0 && cout > 0 && cout % 64 == 0, HSEFR_ERR_UNSUPPORTED,
                  "conv1x1_proj_bf16: channels %d / %d -> %d must be multiples of 64", K, k2, cout);
    HSEFR_REQUIRE(n >= 0 && oh > 0 && ow > 0 && stride >= 1 && (oh - 1) * stride < h2 && (ow - 1) * stride < w2, HSEFR_ERR_INVALID,
                  "conv1x1_proj_bf16: a %dx%d output is not a stride-%d view of a %dx%d input", oh, ow, stride, h2, w2);
    HSEFR_REQUIRE(act == HSEFR_ACT_NONE || act == HSEFR_ACT_RELU || act == HSEFR_ACT_RELU6, HSEFR_ERR_UNSUPPORTED, "conv1x1_proj_bf16: act %d", act);
    if (n == 0) return HSEFR_OK;
    const long long P = (long long)n * oh * ow;
    if (conv1x1_w4_proj_preferred(P, K, k2, cout) && conv1x1_w4_bf16_supported(n, oh, ow, K, oh, ow, cout, 1) &&
        conv1x1_w4_bf16_supported(n, h2, w2, k2, oh, ow, cout, stride))      // the matrix-bound pairs: four wide MFMA waves + loaders (csrc/conv1x1_w4_bf16.hip)
        return launch_conv1x1_w4_proj_bf16(x, wt, scale, shift, x2, wt2, scale2, shift2, y, n, oh, ow, K, cout, k2, stride, h2, w2, act, s);
    ProjParams pj{};
    pj.x2 = (const u16*)x2; pj.wt2 = (const u16*)wt2; pj.scale2 = scale2; pj.shift2 = shift2;
    pj.K2 = k2; pj.stride = stride; pj.H2 = h2; pj.W2 = w2; pj.OH = oh; pj.OW = ow;
    pj.x2_bytes = (long long)n * h2 * w2 * k2 * 2;
    HSEFR_REQUIRE(P < (1ll << 31) && pj.x2_bytes < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv1x1_proj_bf16: tensors beyond 2 GB");
    const long long t128 = ((P + 127) / 128) * (cout / 128);
    if (cout % 128 == 0 && t128 >= 768)
        return launch_proj_cfg<128, 128, 2>((const u16*)x, (const u16*)wt, scale, shift, (u16*)y, P, K, cout, act, pj, s);
    return launch_proj_cfg<128, 64, 3>((const u16*)x, (const u16*)wt, scale, shift, (u16*)y, P, K, cout, act, pj, s);
}

}  // namespace hsefr
