// bf16 convolution (1x1 / 3x3 / any KH x KW, stride, zero padding) + scale + shift (+ residual) + act as an implicit GEMM whose
// BOTH operands reach LDS by LDS-DMA -- the ResNet-50 trunk (resnet50_ft, the graph behind vgg2_resnet.pb at facerec_test.py:213)
// on the skeleton of pwconv_ps.hip.  NHWC bf16 in, bf16 out, fp32 accumulation, gfx950.
//
//   Y[p, n] = act( bf16( scale[n] * sum_{kh,kw,c} X[pix(p) + (kh,kw), c] * Wt[n, (kh*KW+kw)*C + c] + shift[n] ) (+ R[p, n]) )
//                                                                   (the rounding points of conv_bf16.hip / oracle/resnet50.py)
//
// Why a third bf16 kernel: conv_bf16.hip and conv1x1_bf16.hip stage both operands global -> VGPR -> ds_write_b128.  A 16-byte
// LDS store moves 1 KiB per wave in ~13 cycles (79 B/clk/CU) against 256 B/clk for the fragment reads; per 64-deep K-step of a
// 128 x 128 tile that is ~415 cycles of stores + 256 of reads under 512 cycles of MFMA: the LDS port, not the matrix pipe, set the
// pace (335-450 TFLOP/s measured on the 3x3 layers).  `buffer_load_dwordx4 ... lds` writes the tile without the VGPR hop and without
// the ds_write: per K-step the LDS sees only the fragment reads.
//
//   * the im2col GATHER is done by the DMA's per-lane source address: a piece = 8 output pixels x 128 B (64 channels of one tap);
//     lane (pixel, 16-B chunk) points at input pixel (oh*s - pad + kh, ow*s - pad + kw), chunk permuted so that the linear LDS
//     image is the swizzled one; a padding tap, or a row beyond the last pixel, gets an offset outside the buffer resource and
//     the hardware writes ZEROS for it (no branches, no masks in the MFMA waves);
//   * one persistent workgroup of 12 waves per CU: 8 MFMA waves + 4 loader waves that issue every piece, three LDS stages, the DMA
//     two K-steps ahead (pwconv_ps.hip's protocol: one barrier per step, one more per tile while the MFMA waves store);
//   * v_mfma_f32_16x16x32_bf16 with the weights as the first operand; weight rows are read from LDS in a permuted order so that a
//     lane's two channel blocks hold 8 CONSECUTIVE output channels of one pixel: the tile leaves the accumulators as 16-byte
//     stores (16 pixels x 64 contiguous bytes per instruction), the residual arrives the same way;
//   * tile shapes: (32 RB) x 128 with MFMA waves 2 x 4, RB = 4..9, for Cout % 128 == 0; (64 RB) x 64 with waves 4 x 2, RB = 2..5,
//     for Cout = 64; RB is picked per layer so that the tiles fill whole rounds of 256 workgroups (ResNet-50 at batch 128 has
//     M = 49 * 2^k pixels: 25088 x 256 is 224 tiles of 224 x 128, one round on 7/8 of the chip).
// Every output element is accumulated over K in one fixed order by one wave: bit-identical run to run, independent of the grid.
#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int ROWB = 128;       // bytes per LDS row: 64 bf16 = one K-step

#ifdef HSEFR_CD_STAMPS
// Diagnostic build only (HSEFR_DEV=1 HSEFR_EXTRA_FLAGS=-DHSEFR_CD_STAMPS build.sh): per-wave s_memtime sums of the step phases.
__device__ unsigned long long g_cd_stamps[256 * 12 * 8];
#define CD_STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); st[i] += _t - tprev; tprev = _t; } while (0)
#define CD_STAMP_DECL unsigned long long st[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tprev = __builtin_amdgcn_s_memtime(); const unsigned long long tstart = tprev
#define CD_STAMP_FLUSH do { if (lane == 0 && blockIdx.x < 256) { unsigned long long* o = g_cd_stamps + (blockIdx.x * 12 + wave) * 8; \
    for (int i_ = 0; i_ < 6; ++i_) o[i_] = st[i_]; o[6] = __builtin_amdgcn_s_memtime() - tstart; o[7] = nsteps; } } while (0)
#else
#define CD_STAMP(i) do { } while (0)
#define CD_STAMP_DECL do { } while (0)
#define CD_STAMP_FLUSH do { } while (0)
#endif

__device__ __forceinline__ int swz_key(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 2); }
__device__ __forceinline__ unsigned f2bf_bits(float f) { return hsefr_bf16_bits(f); }      // round-to-nearest-even (common.h)
__device__ __forceinline__ float bfround(float f) { return __uint_as_float(f2bf_bits(f) << 16); }

// A buffer resource whose words are pinned to SGPRs: the inline-asm DMA / store take it under an "s" constraint, and with the
// parameter block behind by-reference lambdas hipcc otherwise keeps (selects between) resources in VGPRs -- which assembles to
// an invalid instruction, not to a waterfall loop.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_sgpr(const void* ptr, long long bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned n = __builtin_amdgcn_readfirstlane(bytes <= 0 ? 0u : (bytes > 0xffffffffll ? 0xffffffffu : (unsigned)bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}

struct ConvDmaParams {
    const void* x;       // [N,H,W,C] bf16
    const void* wt;      // [Cout][KH*KW*C] bf16
    const float* scale;  // [Cout]
    const float* shift;  // [Cout]
    const void* res;     // [M,Cout] bf16 or null
    void* y;             // [M,Cout] bf16
    long long x_bytes;
    int H, W, C, OH, OW, Cout, KH, KW, stride, pad_t, pad_l;
    float act_lo, act_hi;   // clamp bounds: (-inf, +inf) none, (0, +inf) ReLU, (0, 6) ReLU6
    unsigned M;             // N*OH*OW output pixels
    unsigned tiles_n, total_tiles;
    int reverse;
};

template <int RB, int WAVES_M>
__global__ __launch_bounds__(768, 1) void conv_dma_bf16_kernel(ConvDmaParams p) {
    constexpr int WAVES_N = 8 / WAVES_M;
    constexpr int BM = WAVES_M * 16 * RB, BN = WAVES_N * 32;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int NPIECE = (BM + BN) / 8;
    constexpr int APW = BM / 32, BPW = BN / 32;     // pieces per loader wave and K-step: activation (gathered) | weight
    constexpr int PPW = APW + BPW;
    static_assert(NPIECE == 4 * PPW, "pieces divide over the four loader waves");
    constexpr int E_OFF = 3 * STAGE;               // epilogue constants by tile parity: [scale piece 1 KiB | shift piece 1 KiB] x 2
    static_assert(E_OFF + 4096 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[E_OFF + 4096];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int CS = p.C / 64;                        // channel slabs per tap
    const int KT = p.KH * p.KW * CS;
    if (blockIdx.x >= p.total_tiles) return;
    const unsigned ntile = (p.total_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const unsigned nsteps = ntile * KT;
    const unsigned wrowbytes = (unsigned)KT * 128u;

    auto tile_origin = [&](unsigned i, unsigned& mm0, int& nn0) __attribute__((always_inline)) {
        const unsigned lt = xcd_remap_dir(blockIdx.x + (i < ntile ? i : ntile - 1) * gridDim.x, p.total_tiles, p.reverse);
        const unsigned tm = lt / p.tiles_n;
        mm0 = tm * BM;
        nn0 = (int)(lt - tm * p.tiles_n) * BN;
    };
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

    if (wave >= 8) {
        // =================================== loader waves 8..11 ===================================
        const int lw = wave - 8;
        const char* w_ptr = nullptr;      // weight rows of the prefetch cursor's tile
        long long w_bytes = 0;
        // Every loader wave carries APW gathered activation pieces and BPW weight pieces per step (compile-time roles: the issue
        // loop is branch-free).  Per activation piece: byte offset of the lane's 16 B for tap (0, 0), slab 0, and the tap validity bits.
        unsigned pbase[APW], pmask[APW], pvb[BPW];
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int r = (lw * BPW + j) * 8 + (lane >> 3);
            pvb[j] = (unsigned)r * wrowbytes + 16u * (unsigned)((lane & 7) ^ swz_key(r));
        }
        auto piece = [&](const __amdgpu_buffer_rsrc_t& r, unsigned lds_addr, unsigned voff, unsigned soff) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff),
                         "s"(r), "s"(__builtin_amdgcn_readfirstlane(soff))
                         : "memory", "m0");
        };
        unsigned pf_i = 0, pf_step = 0;
        int pf_kt = 0, pf_tap = 0, pf_cs = 0, pf_kh = 0, pf_kw = 0, pf_n0 = 0;
        const unsigned ohow = (unsigned)(p.OH * p.OW);
        auto setup_tile = [&](unsigned i) __attribute__((always_inline)) {
            unsigned mm0;
            tile_origin(i, mm0, pf_n0);
            w_ptr = (const char*)p.wt + (long long)pf_n0 * wrowbytes;
            w_bytes = (long long)(p.Cout - pf_n0) * wrowbytes;
#pragma unroll
            for (int j = 0; j < APW; ++j) {
                const int r = (lw * APW + j) * 8 + (lane >> 3);
                const unsigned chunk = (unsigned)((lane & 7) ^ swz_key(r));
                const unsigned m = mm0 + (unsigned)r;
                const unsigned mc = m < p.M ? m : p.M - 1u;
                const unsigned n = mc / ohow, rem = mc - n * ohow;
                const unsigned oh = rem / (unsigned)p.OW, ow = rem - oh * (unsigned)p.OW;
                const int ih0 = (int)oh * p.stride - p.pad_t, iw0 = (int)ow * p.stride - p.pad_l;
                pbase[j] = (unsigned)(((int)(n * (unsigned)p.H) + ih0) * p.W + iw0) * (unsigned)(p.C * 2) + 16u * chunk;
                unsigned mk = 0;
                for (int kh = 0, t = 0; kh < p.KH; ++kh)
                    for (int kw = 0; kw < p.KW; ++kw, ++t) {
                        const bool ok = (unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W;
                        mk |= (ok ? 1u : 0u) << t;
                    }
                pmask[j] = m < p.M ? mk : 0u;
            }
        };
        auto issue_step = [&]() __attribute__((always_inline)) {
            const unsigned base = lds0 + (pf_step % 3u) * STAGE;
            // K order: channel slab OUTER, tap INNER -- the KH x KW taps of one 64-channel slab read (nearly) the same input pixels
            // in consecutive steps, so all but the first come out of L2 (tap-outer order re-read each pixel C / 64 steps later,
            // ~1 MB of other traffic per step and XCD in between: the 4 MB L2 did not hold it and the loads ran at MALL speed)
            const unsigned a_adv = (unsigned)((pf_kh * p.W + pf_kw) * p.C * 2 + pf_cs * 128);
            const unsigned b_adv = (unsigned)(pf_tap * CS + pf_cs) * 128u;
            const __amdgpu_buffer_rsrc_t rx = make_rsrc_sgpr(p.x, p.x_bytes), rw = make_rsrc_sgpr(w_ptr, w_bytes);
            // (offsets in registers of their own, all computed before the first piece goes out; pieces spread over 8 L2 channels
            // instead of 8 neighbouring pixels were tried: no effect on the issue rate)
            unsigned voff[PPW];
#pragma unroll
            for (int j = 0; j < APW; ++j) voff[j] = ((pmask[j] >> pf_tap) & 1u) ? pbase[j] + a_adv : 0x80000000u;
#pragma unroll
            for (int j = 0; j < BPW; ++j) voff[APW + j] = pvb[j] + b_adv;
#pragma unroll
            for (int j = 0; j < PPW; ++j) asm volatile("" : "+v"(voff[j]));
#pragma unroll
            for (int j = 0; j < APW; ++j) piece(rx, base + (lw * APW + j) * 1024, voff[j], 0u);
#pragma unroll
            for (int j = 0; j < BPW; ++j) piece(rw, base + BM * ROWB + (lw * BPW + j) * 1024, voff[APW + j], 0u);
            ++pf_step;
            ++pf_kt;
            ++pf_tap;
            if (++pf_kw == p.KW) {
                pf_kw = 0;
                if (++pf_kh == p.KH) { pf_kh = 0; pf_tap = 0; ++pf_cs; }
            }
            if (pf_kt == KT) {
                pf_kt = pf_tap = pf_cs = pf_kh = pf_kw = 0;
                setup_tile(++pf_i);
            }
        };
        setup_tile(0);
        int e_n0 = pf_n0;
        unsigned ci = 0;
        int ckt = 0;
        // k counts ISSUED steps: step k goes out, then (k >= 1) step k - 1 is waited for and handed over at the barrier; the MFMA
        // waves work on step k - 2.  One call site of issue_step: the per-tile address decode is inlined once.
        CD_STAMP_DECL;
        for (unsigned k = 0; k < nsteps + 2; ++k) {
            if (k >= 2 && ckt == 0 && lw == 3) {
                // the tile's epilogue constants by LDS-DMA: scale[n0 .. n0 + 127] as lanes 0-31 of one piece, shift[..] as lanes
                // 32-63 of a second one; issued AHEAD of this step's pieces so the counted wait covers them; two copies by tile parity
                const __amdgpu_buffer_rsrc_t rd = make_rsrc_sgpr(p.scale + e_n0, (long long)(p.Cout - e_n0) * 4),
                                             rs = make_rsrc_sgpr(p.shift + e_n0, (long long)(p.Cout - e_n0) * 4);
                const unsigned eb = lds0 + E_OFF + (ci & 1u) * 2048u;
                piece(rd, eb, lane < 32 ? 16u * lane : 0x80000000u, 0u);
                piece(rs, eb + 1024, lane >= 32 ? 16u * (unsigned)(lane - 32) : 0x80000000u, 0u);
            }
            issue_step();
            CD_STAMP(0);
            if (k == 0) continue;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");   // step k - 1 (and the constants) have landed: vmcnt retires in order
            CD_STAMP(1);
            __syncthreads();
            CD_STAMP(2);
            if (k >= 2 && ++ckt == KT) {
                ckt = 0;
                unsigned mm0;
                tile_origin(++ci, mm0, e_n0);
                __syncthreads();                            // pause while the MFMA waves store the tile
                CD_STAMP(3);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CD_STAMP_FLUSH;
        return;
    }

    // =================================== MFMA waves 0..7: wave tile rows [wm * 16 RB, +16 RB) x channels [wn * 32, +32) ===================================
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l16 = lane & 15, lq = lane >> 4;
    const int arow = wm * 16 * RB + l16;
    const int a_g0 = arow * ROWB + 16 * (lq ^ swz_key(arow)), a_g1 = arow * ROWB + 16 * ((4 + lq) ^ swz_key(arow));
    // weight rows in permuted order: MFMA row i of channel block nb is channel 8 (i >> 2) + 4 nb + (i & 3) of the wave's 32
    int b_g0[2], b_g1[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int brow = BM + wn * 32 + 8 * (l16 >> 2) + 4 * nb + (l16 & 3);
        b_g0[nb] = brow * ROWB + 16 * (lq ^ swz_key(brow));
        b_g1[nb] = brow * ROWB + 16 * ((4 + lq) ^ swz_key(brow));
    }
    f32x4 acc[RB][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    // epilogue geometry: one pixel per lane and 16-row block, channels wn * 32 + 8 lq .. + 7: 16 bytes
    const unsigned yvoff = ((unsigned)(wm * 16 * RB + l16) * (unsigned)p.Cout + (unsigned)(wn * 32 + 8 * lq)) * 2u;

    unsigned m0;
    int n0;
    unsigned ci = 0;
    int ckt = 0;
    tile_origin(0, m0, n0);
    __syncthreads();                                        // (the loaders' prologue barrier: step 0 has landed)
    CD_STAMP_DECL;

    constexpr int HB = RB >= 4 ? 2 : 1;                     // row blocks whose MFMAs are held back behind the step barrier
    const bf16x8 fzero = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 h_a0[HB], h_a1[HB], h_b00 = fzero, h_b01 = fzero, h_b10 = fzero, h_b11 = fzero;
#pragma unroll
    for (int i = 0; i < HB; ++i) h_a0[i] = h_a1[i] = fzero;
    auto mfma_block = [&](int rb, const bf16x8& x0, const bf16x8& x1, const bf16x8& w00, const bf16x8& w01, const bf16x8& w10, const bf16x8& w11) {
        acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w00, x0, acc[rb][0], 0, 0, 0);
        acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w10, x0, acc[rb][1], 0, 0, 0);
        acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w01, x1, acc[rb][0], 0, 0, 0);
        acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w11, x1, acc[rb][1], 0, 0, 0);
    };

    for (unsigned g = 0; g < nsteps; ++g) {
        const unsigned char* stg = smem + (g % 3u) * STAGE;
        const bf16x8 b00 = *(const bf16x8*)(stg + b_g0[0]), b01 = *(const bf16x8*)(stg + b_g1[0]);
        const bf16x8 b10 = *(const bf16x8*)(stg + b_g0[1]), b11 = *(const bf16x8*)(stg + b_g1[1]);
        bf16x8 a0[RB], a1[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            a0[rb] = *(const bf16x8*)(stg + a_g0 + rb * 16 * ROWB);
            a1[rb] = *(const bf16x8*)(stg + a_g1 + rb * 16 * ROWB);
        }
#pragma unroll
        for (int i = 0; i < HB; ++i) mfma_block(RB - HB + i, h_a0[i], h_a1[i], h_b00, h_b01, h_b10, h_b11);
#pragma unroll
        for (int rb = 0; rb < RB - HB; ++rb) mfma_block(rb, a0[rb], a1[rb], b00, b01, b10, b11);
        constexpr int PRE = RB >= 2 ? 2 : 1;
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * PRE, 0);
#pragma unroll
        for (int grp = 0; grp < RB; ++grp) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (grp + PRE < RB) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
#pragma unroll
        for (int i = 0; i < HB; ++i) { h_a0[i] = a0[RB - HB + i]; h_a1[i] = a1[RB - HB + i]; }
        h_b00 = b00; h_b01 = b01; h_b10 = b10; h_b11 = b11;
        CD_STAMP(0);
        __syncthreads();                                // step g + 1 has landed; slot g % 3 is released
        CD_STAMP(1);
        if (++ckt == KT) {
#pragma unroll
            for (int i = 0; i < HB; ++i) mfma_block(RB - HB + i, h_a0[i], h_a1[i], h_b00, h_b01, h_b10, h_b11);
#pragma unroll
            for (int i = 0; i < HB; ++i) h_a0[i] = h_a1[i] = fzero;
            h_b00 = h_b01 = h_b10 = h_b11 = fzero;
            const long long yorg = ((long long)m0 * p.Cout + n0) * 2ll, ybytes = ((long long)(p.M - m0) * p.Cout - n0) * 2ll;
            const __amdgpu_buffer_rsrc_t ry = make_rsrc_sgpr((char*)p.y + yorg, ybytes);
            const __amdgpu_buffer_rsrc_t rr = make_rsrc_sgpr((const char*)p.res + yorg, p.res ? ybytes : 0);
            f32x4 rres[RB];
            if (p.res) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    rres[rb] = bload16(rr, yvoff, __builtin_amdgcn_readfirstlane((unsigned)(16 * rb) * (unsigned)p.Cout * 2u));
            }
            f32x4 e_sc[2], e_sh[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                e_sc[nb] = *(const f32x4*)(smem + E_OFF + (ci & 1u) * 2048 + (wn * 32 + 8 * lq + 4 * nb) * 4);
                e_sh[nb] = *(const f32x4*)(smem + E_OFF + (ci & 1u) * 2048 + 1024 + 512 + (wn * 32 + 8 * lq + 4 * nb) * 4);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float v[8];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * nb + e] = fmaf(acc[rb][nb][e], e_sc[nb][e], e_sh[nb][e]);
                if (p.res) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const unsigned rw2 = __float_as_uint(rres[rb][d]);
                        v[2 * d] = bfround(v[2 * d]) + __uint_as_float(rw2 << 16);
                        v[2 * d + 1] = bfround(v[2 * d + 1]) + __uint_as_float(rw2 & 0xFFFF0000u);
                    }
                }
                f32x4 o;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float f0 = fminf(fmaxf(v[2 * d], p.act_lo), p.act_hi), f1 = fminf(fmaxf(v[2 * d + 1], p.act_lo), p.act_hi);
                    o[d] = __uint_as_float(hsefr_pack_bf16x2(f0, f1));
                }
                bstore16_welded(o, ry, yvoff, __builtin_amdgcn_readfirstlane((unsigned)(16 * rb) * (unsigned)p.Cout * 2u));
            }
            zero_acc();
            ckt = 0;
            tile_origin(++ci, m0, n0);
            CD_STAMP(2);
            __syncthreads();                            // lets the loaders go on
            CD_STAMP(3);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // stores issued from asm: drained before the wave ends
    CD_STAMP_FLUSH;
}

struct DmaCfg { int waves_m, rb; };

// Tile shape per layer: fewest (rounds of 256 workgroups) x (rows per tile + a fixed per-tile cost) x (the tile's width).  Layers whose
// channel count allows the 128-wide tiles may still take the 64-wide ones: a stride-2 3x3 on a 14 x 14 map at batch 128 is 6272 rows --
// 98 tiles of 128 x 128 on 256 CUs; as 196 tiles of 128 x 64 it is one round of tiles that cost 0.6 of a wide one (32 -> 20 us)
DmaCfg choose_cfg(long long m, int cout, int forced_rb) {
    DmaCfg best{0, 0};
    double best_cost = -1;
    for (int wide = cout % 128 == 0 ? 1 : 0; wide >= 0; --wide) {
        const int wavesm = wide ? 2 : 4, bn = wide ? 128 : 64;
        const int rb_lo = wide ? 4 : 2, rb_hi = wide ? 9 : 5;
        for (int rb = rb_hi; rb >= rb_lo; --rb) {
            if (forced_rb > 0 && rb != forced_rb) continue;
            const long long bm = (long long)wavesm * 16 * rb;
            const long long tiles = ((m + bm - 1) / bm) * (cout / bn);
            const double cost = (double)((tiles + 255) / 256) * ((double)bm + 48.0) * (wide ? 1.0 : 0.6);
            if (best_cost < 0 || cost < best_cost - 1e-9) { best = DmaCfg{wavesm, rb}; best_cost = cost; }
        }
    }
    return best;
}

HSEFR_KNOB(g_cd_rb, 0);     // dev builds: forced RB
HSEFR_KNOB(g_cd_off, 0);    // dev builds: 1 = never use this kernel, 2 = use it for every shape it covers (A/B timing)

template <int RB, int WAVES_M>
int launch_cfg(ConvDmaParams& p, hipStream_t s) {
    constexpr int BM = WAVES_M * 16 * RB, BN = (8 / WAVES_M) * 32;
    const long long tiles_m = ((long long)p.M + BM - 1) / BM;
    p.tiles_n = (unsigned)(p.Cout / BN);
    const long long total = tiles_m * p.tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv_dma_bf16: too many tiles");
    p.total_tiles = (unsigned)total;
    const unsigned g = (unsigned)(total < 256 ? total : 256);
    HSEFR_LAUNCH((conv_dma_bf16_kernel<RB, WAVES_M>), dim3(g), dim3(768), 0, s, p);
    return launch_status("conv_dma_bf16");
}

}  // namespace

#ifdef HSEFR_DEV
int read_cd_stamps(void* host_out, size_t bytes) {
#ifdef HSEFR_CD_STAMPS
    HSEFR_REQUIRE(bytes <= sizeof(unsigned long long) * 256 * 12 * 8, HSEFR_ERR_INVALID, "read_cd_stamps: too many bytes");
    HSEFR_HIP_CHECK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_cd_stamps), bytes));
    return HSEFR_OK;
#else
    (void)host_out; (void)bytes;
    set_error("read_cd_stamps: library built without -DHSEFR_CD_STAMPS");
    return HSEFR_ERR_UNSUPPORTED;
#endif
}
void set_cd_rb(int v) { g_cd_rb = v; }
void set_cd_off(int v) { g_cd_off = v; }
#endif

bool conv_dma_forced() { return g_cd_off == 2; }

bool conv_dma_bf16_supported(long long n, int h, int w, int c, int oh, int ow, int cout, int kh, int kw) {
    if (g_cd_off == 1) return false;
    const long long xbytes = n * h * w * (long long)c * 2, ybytes = n * oh * ow * (long long)cout * 2;
    return c > 0 && c % 64 == 0 && cout > 0 && cout % 64 == 0 && kh > 0 && kw > 0 && kh * kw <= 32 && xbytes < (1ll << 31) &&
           ybytes < (1ll << 32) && n * oh * ow < (1ll << 31) && 640ll * cout * 2 < (1ll << 31) && (long long)kh * kw * c * 2 * 128 < (1ll << 31);
}

int launch_conv_dma_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y, int n, int h,
                         int w, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, int act, hipStream_t s) {
    HSEFR_REQUIRE(conv_dma_bf16_supported(n, h, w, c, oh, ow, cout, kh, kw), HSEFR_ERR_UNSUPPORTED, "conv_dma_bf16: shape not covered");
    if (n == 0) return HSEFR_OK;
    ConvDmaParams p;
    p.x = x; p.wt = wt; p.scale = scale; p.shift = shift; p.res = res; p.y = y;
    p.x_bytes = (long long)n * h * w * c * 2;
    p.H = h; p.W = w; p.C = c; p.OH = oh; p.OW = ow; p.Cout = cout; p.KH = kh; p.KW = kw; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
    p.act_lo = act == HSEFR_ACT_NONE ? -INFINITY : 0.f;
    p.act_hi = act == HSEFR_ACT_RELU6 ? 6.f : INFINITY;
    HSEFR_REQUIRE(act == HSEFR_ACT_NONE || act == HSEFR_ACT_RELU || act == HSEFR_ACT_RELU6, HSEFR_ERR_UNSUPPORTED, "conv_dma_bf16: act %d", act);
    p.M = (unsigned)((long long)n * oh * ow);
    p.reverse = sweep_reverse();
    const DmaCfg cfg = choose_cfg(p.M, cout, g_cd_rb);
    HSEFR_REQUIRE(cfg.rb > 0, HSEFR_ERR_UNSUPPORTED, "conv_dma_bf16: no tile shape (forced RB %d)", (int)g_cd_rb);
    if (cfg.waves_m == 2) {
        switch (cfg.rb) {
            case 4: return launch_cfg<4, 2>(p, s);
            case 5: return launch_cfg<5, 2>(p, s);
            case 6: return launch_cfg<6, 2>(p, s);
            case 7: return launch_cfg<7, 2>(p, s);
            case 8: return launch_cfg<8, 2>(p, s);
            default: return launch_cfg<9, 2>(p, s);
        }
    }
    switch (cfg.rb) {
        case 2: return launch_cfg<2, 4>(p, s);
        case 3: return launch_cfg<3, 4>(p, s);
        case 4: return launch_cfg<4, 4>(p, s);
        default: return launch_cfg<5, 4>(p, s);
    }
}

}  // namespace hsefr
