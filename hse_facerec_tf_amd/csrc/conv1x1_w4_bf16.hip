// 1x1 bf16 convolution (stride 1 or 2) + scale + shift (+ residual | + projected shortcut) + act as a GEMM with FOUR wide MFMA
// waves and four LDS-DMA loader waves (round 5): the "reduce" / "increase" / projection layers of ResNet-50's bottlenecks
// (resnet50_ft, the graph behind vgg2_resnet.pb at facerec_test.py:213), two thirds of that network's time once its 3x3 layers run
// on conv3x3_w2_bf16.hip.  NHWC bf16 in / out, fp32 accumulation, gfx950; rounding points of oracle/resnet50.py:
//
//   Y[p, n] = act( bf16( scale[n] * sum_c X[pix(p), c] * Wt[n, c] + shift[n] ) (+ R[p, n]) )
//
// What the stamps of round 5 said about the kernels these layers ran on (profiles/r05_conv_stamps.txt): conv1x1_bf16.hip
// (register-staged, 128 x 128 tiles, 4 + 4 waves per CU) is EPILOGUE-bound on the short-K increase layers -- scale / shift /
// bf16 rounding / LDS transpose / residual / second rounding: ~600 vector instructions per thread and tile around four K-steps --
// and conv_dma_bf16.hip (8 MFMA waves of 112 x 32) spends 51 % of an increase layer in its epilogue while its loaders wait.  Here:
//   * 4 MFMA waves with (16 RB) x 64 wave tiles (224 x 128 as 2 x 2, or 256 x 64 as 4 x 1 for the 64-channel layers) + 4 loader
//     waves: 256 registers per wave -- 112 accumulators AND the tile's 14 residual fragments, which are requested at the top of the
//     tile's LAST K-step and are there when the epilogue starts;
//   * weights first in v_mfma_f32_16x16x32_bf16 with permuted weight rows: a lane owns 8 consecutive channels, the tile leaves the
//     accumulators as 16-byte stores (no LDS transpose), rounding on v_cvt_pk_bf16_f32;
//   * separate rings: THREE activation stages (28 KB each), FOUR weight stages (16 KB) -- the weights are published one step early,
//     a step's last row block re-loads each weight fragment for the next step right after its last use; the loaders never pause:
//     while the MFMA waves store a tile the next tile's first K-steps land (for K <= 192 the whole K loop);
//   * activation rows are GATHERED by the DMA's per-lane source address: a stride-2 layer costs the same as a dense one;
//   * swizzle chunk ^ (row & 6): conflict-free ds_read_b128 for any 16 consecutive rows (conv3x3_w2_bf16.hip).
// Every output element is accumulated over K in one fixed order by one wave: bit-identical run to run, independent of the grid.
//
// Where it runs (conv1x1_w4_bf16_preferred / conv1x1_w4_proj_preferred, measured in the network at batch 128): the K-deep REDUCTIONS of the
// 28- and 14-pixel stages, stride 1 and 2 (26 -> 23 us per layer); PROJ -- the increase layer of a stage's first block with its
// projected shortcut as extra K-steps in front (conv1x1_bf16.hip's header: same rounding points) -- on the 28-, 14- and 7-pixel stages,
// where the pair is 39.5 GFLOP of matrix work (78 -> 64, 74 -> 54, 72 -> 48 us); the K = 512 increase layers of the 7-pixel stage.  The
// other increase layers (K <= 256) are 10 % slower here than on the register-staged kernel's two workgroups per CU: they are bound
// by what the CU's vector-memory path moves (DESIGN.md lesson 54).
#include <type_traits>

#include "common.h"

#ifndef W4_RESAUX
#define W4_RESAUX 0     // cache policy of the residual loads (buffer aux bits: 1 glc, 2 slc), as conv1x1_bf16.hip's C11_RESAUX
#endif
namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ROWB = 128;                 // bytes per LDS row: 64 bf16 = one K-step
constexpr int NA = 3, NB = 4;             // ring stages: activations, weights

#ifdef HSEFR_CD_STAMPS
__device__ unsigned long long g_w4_stamps[256 * 8 * 8];
#define W4_STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); st[i] += _t - tprev; tprev = _t; } while (0)
#define W4_STAMP_DECL unsigned long long st[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tprev = __builtin_amdgcn_s_memtime(); const unsigned long long tstart = tprev
#define W4_STAMP_FLUSH do { if (lane == 0 && blockIdx.x < 256) { unsigned long long* o = g_w4_stamps + (blockIdx.x * 8 + wave) * 8; \
    for (int i_ = 0; i_ < 6; ++i_) o[i_] = st[i_]; o[6] = __builtin_amdgcn_s_memtime() - tstart; o[7] = nsteps; } } while (0)
#else
#define W4_STAMP(i) do { } while (0)
#define W4_STAMP_DECL do { } while (0)
#define W4_STAMP_FLUSH do { } while (0)
#endif

__device__ __forceinline__ float bfround(float f) { return __uint_as_float(hsefr_bf16_bits(f) << 16); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_sgpr(const void* ptr, long long bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned n = __builtin_amdgcn_readfirstlane(bytes <= 0 ? 0u : (bytes > 0xffffffffll ? 0xffffffffu : (unsigned)bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}
__device__ __forceinline__ void step_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct W4Params {
    const void* x;       // [N,H,W,K] bf16
    const void* wt;      // [Cout][K] bf16
    const float* scale;
    const float* shift;
    const void* res;     // [M,Cout] bf16 or null
    void* y;             // [M,Cout] bf16
    long long x_bytes;
    int K, Cout;
    int stride, H, W, OH, OW;       // output pixel (n, oh, ow) reads input pixel (n, oh * stride, ow * stride)
    float act_lo, act_hi;
    unsigned M;                     // N * OH * OW
    unsigned tiles_n, total_tiles;
    int reverse;
    // PROJ kernels (the projected shortcut of a stage's first block, conv1x1_bf16.hip's header): a tile's K loop starts with the K2 / 64
    // steps of x2[pixel * stride2, :] . wt2, whose result -- scale2, shift2, rounded to bf16 -- takes the residual's place
    const void* x2;      // [N,H2,W2,K2] bf16: the block's input
    const void* wt2;     // [Cout][K2] bf16
    const float* scale2;
    const float* shift2;
    long long x2_bytes;
    int K2, stride2, H2, W2;
    int b_resident;      // 1: the K loop's steps divide the four weight stages (K = 64, 128 or 256, no projection) AND every tile of a workgroup has
                         // the same channel origin -- stage s then always holds the SAME weight rows: they are loaded with the first four
                         // steps and never again (16 of a step's 44 KB on layers that are bound by what the CU's loaders can issue)
};

template <int RB, int WAVES_M, bool PROJ = false>
__global__ __launch_bounds__(512) void conv1x1_w4_bf16_kernel(W4Params p) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int BM = WAVES_M * RB * 16, BN = WAVES_N * 64;
    constexpr int APW = BM / 32, BPW = BN / 32;     // pieces per loader wave and step: activation rows | weight rows
    static_assert(BM % 32 == 0, "activation pieces divide over the four loader waves");
    constexpr int ASTAGE = BM * ROWB, BSTAGE = BN * ROWB;
    constexpr int B_OFF = NA * ASTAGE;
    constexpr int E_OFF = B_OFF + NB * BSTAGE;
    constexpr int E_PAR = PROJ ? 4096 : 2048;      // epilogue constants per tile parity: [scale 1 KiB | shift 1 KiB] (| scale2 | shift2)
    static_assert(E_OFF + 2 * E_PAR <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[E_OFF + 2 * E_PAR];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT2 = PROJ ? p.K2 / 64 : 0;          // a tile's K loop: the projection's steps first, then the main product's
    const int KT = p.K / 64 + KT2;
    if (blockIdx.x >= p.total_tiles) return;
    const unsigned ntile = (p.total_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const unsigned nsteps = ntile * (unsigned)KT;
    const unsigned wrowbytes = (unsigned)p.K * 2u;
    const unsigned wrowbytes2 = PROJ ? (unsigned)p.K2 * 2u : 0u;

    auto tile_origin = [&](unsigned i, unsigned& mm0, int& nn0) __attribute__((always_inline)) {
        const unsigned lt = xcd_remap_dir(blockIdx.x + (i < ntile ? i : ntile - 1) * gridDim.x, p.total_tiles, p.reverse);
        const unsigned tm = lt / p.tiles_n;
        mm0 = tm * BM;
        nn0 = (int)(lt - tm * p.tiles_n) * BN;
    };
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

    if (wave >= 4) {
        // =================================== loader waves 4..7 ===================================
        const int lw = wave - 4;
        auto piece = [&](const __amdgpu_buffer_rsrc_t& r, unsigned lds_addr, unsigned voff) __attribute__((always_inline)) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(r)
                         : "memory", "m0");
        };
        const __amdgpu_buffer_rsrc_t rx = make_rsrc_sgpr(p.x, p.x_bytes);
        // weight pieces: LDS row R = 64 wn + 16 nb + i holds output channel 64 wn + 32 (nb >> 1) + 8 (i >> 2) + 4 (nb & 1) + (i & 3)
        unsigned pvb[BPW], pvb2[PROJ ? BPW : 1];
        const __amdgpu_buffer_rsrc_t rx2 = make_rsrc_sgpr(PROJ ? p.x2 : nullptr, PROJ ? p.x2_bytes : 0);
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int R = (lw * BPW + j) * 8 + (lane >> 3);
            const int nb = (R >> 4) & 3, i = R & 15;
            const int ch = (R & ~63) + 32 * (nb >> 1) + 8 * (i >> 2) + 4 * (nb & 1) + (i & 3);
            pvb[j] = (unsigned)ch * wrowbytes + 16u * (unsigned)((lane & 7) ^ (R & 6));
            if (PROJ) pvb2[PROJ ? j : 0] = (unsigned)ch * wrowbytes2 + 16u * (unsigned)((lane & 7) ^ (R & 6));
        }
        // the two cursors: activations two steps ahead of the MFMA waves, weights three
        unsigned pa[APW];                 // activation cursor's tile: byte offset of the lane's 16 bytes, K-tile 0 (out-of-range marker for rows >= M)
        unsigned pa2[PROJ ? APW : 1];     // ... of the projection's gathered input rows
        unsigned a_tile = 0, a_kt = 0, a_step = 0;
        const unsigned ohow = (unsigned)(p.OH * p.OW);
        auto setup_a = [&](unsigned i) __attribute__((always_inline)) {
            unsigned mm0;
            int nn0;
            tile_origin(i, mm0, nn0);
#pragma unroll
            for (int j = 0; j < APW; ++j) {
                const int r = (lw * APW + j) * 8 + (lane >> 3);
                const unsigned chunk = (unsigned)((lane & 7) ^ (r & 6));
                const unsigned m = mm0 + (unsigned)r;
                unsigned pix = m;
                if (p.stride != 1 || p.OH != p.H || p.OW != p.W) {       // (uniform) strided view of the input map
                    const unsigned n = m / ohow, rem = m - n * ohow;
                    const unsigned oh = rem / (unsigned)p.OW, ow = rem - oh * (unsigned)p.OW;
                    pix = (n * (unsigned)p.H + oh * (unsigned)p.stride) * (unsigned)p.W + ow * (unsigned)p.stride;
                }
                pa[j] = m < p.M ? pix * wrowbytes + 16u * chunk : 0x80000000u;
                if (PROJ) {      // output pixel (n, oh, ow) is projected from input pixel (n, oh stride2, ow stride2) of the block's input
                    const unsigned n = m / ohow, rem = m - n * ohow;
                    const unsigned oh = rem / (unsigned)p.OW, ow = rem - oh * (unsigned)p.OW;
                    const unsigned pix2 = (n * (unsigned)p.H2 + oh * (unsigned)p.stride2) * (unsigned)p.W2 + ow * (unsigned)p.stride2;
                    pa2[PROJ ? j : 0] = m < p.M ? pix2 * wrowbytes2 + 16u * chunk : 0x80000000u;
                }
            }
        };
        auto issue_a = [&]() __attribute__((always_inline)) {
            const unsigned base = lds0 + (a_step % NA) * ASTAGE;
            const bool live = a_step < nsteps;
            unsigned voff[APW];
            const bool proj_step = PROJ && a_kt < (unsigned)KT2;       // (wave-uniform)
            if (proj_step) {
#pragma unroll
                for (int j = 0; j < APW; ++j) voff[j] = (live && pa2[PROJ ? j : 0] != 0x80000000u) ? pa2[PROJ ? j : 0] + a_kt * 128u : 0x80000000u;
            } else {
#pragma unroll
                for (int j = 0; j < APW; ++j) voff[j] = (live && pa[j] != 0x80000000u) ? pa[j] + (a_kt - (unsigned)KT2) * 128u : 0x80000000u;
            }
#pragma unroll
            for (int j = 0; j < APW; ++j) asm volatile("" : "+v"(voff[j]));
            if (proj_step) {
#pragma unroll
                for (int j = 0; j < APW; ++j) piece(rx2, base + (lw * APW + j) * 1024, voff[j]);
            } else {
#pragma unroll
                for (int j = 0; j < APW; ++j) piece(rx, base + (lw * APW + j) * 1024, voff[j]);
            }
            ++a_step;
            if (++a_kt == (unsigned)KT) { a_kt = 0; setup_a(++a_tile); }
        };
        const char* w_ptr = nullptr;
        const char* w2_ptr = nullptr;
        long long w_bytes = 0, w2_bytes = 0;
        unsigned b_tile = 0, b_kt = 0, b_step = 0;
        auto setup_b = [&](unsigned i) __attribute__((always_inline)) {
            unsigned mm0;
            int nn0;
            tile_origin(i, mm0, nn0);
            w_ptr = (const char*)p.wt + (long long)nn0 * wrowbytes;
            w_bytes = (long long)(p.Cout - nn0) * wrowbytes;
            if (PROJ) {
                w2_ptr = (const char*)p.wt2 + (long long)nn0 * wrowbytes2;
                w2_bytes = (long long)(p.Cout - nn0) * wrowbytes2;
            }
        };
        auto issue_b = [&]() __attribute__((always_inline)) {
            const bool proj_step = PROJ && b_kt < (unsigned)KT2;       // (wave-uniform)
            const __amdgpu_buffer_rsrc_t rw = make_rsrc_sgpr(proj_step ? w2_ptr : w_ptr, proj_step ? w2_bytes : w_bytes);
            const unsigned base = lds0 + B_OFF + (b_step & (NB - 1)) * BSTAGE;
            const bool live = b_step < nsteps;
            if (!(p.b_resident && b_step >= (unsigned)NB)) {      // (wave-uniform; resident weights: the stage already holds these rows)
                unsigned voff[BPW];
#pragma unroll
                for (int j = 0; j < BPW; ++j)
                    voff[j] = !live ? 0x80000000u : proj_step ? pvb2[PROJ ? j : 0] + b_kt * 128u : pvb[j] + (b_kt - (unsigned)KT2) * 128u;
#pragma unroll
                for (int j = 0; j < BPW; ++j) asm volatile("" : "+v"(voff[j]));
#pragma unroll
                for (int j = 0; j < BPW; ++j) piece(rw, base + (lw * BPW + j) * 1024, voff[j]);
            }
            ++b_step;
            if (++b_kt == (unsigned)KT) { b_kt = 0; setup_b(++b_tile); }
        };

        // ---- prologue: activations of steps 0, 1; weights of steps 0, 1, 2 ----
        setup_a(0);
        setup_b(0);
        issue_a();
        issue_b();
        issue_a();
        issue_b();
        issue_b();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        step_barrier();                                        // P
        unsigned ci = 0, ckt = 0;
        W4_STAMP_DECL;
        for (unsigned g = 0; g < nsteps; ++g) {
            if (ckt == 0 && lw == 3) {
                // the tile's epilogue constants by LDS-DMA (scale as lanes 0-31 of one piece, shift as lanes 32-63 of a second one),
                // ahead of this step's pieces: the counted wait of the NEXT step covers them
                unsigned mm0;
                int e_n0;
                tile_origin(ci, mm0, e_n0);
                const __amdgpu_buffer_rsrc_t rd = make_rsrc_sgpr(p.scale + e_n0, (long long)(p.Cout - e_n0) * 4),
                                             rs = make_rsrc_sgpr(p.shift + e_n0, (long long)(p.Cout - e_n0) * 4);
                const unsigned eb = lds0 + E_OFF + (ci & 1u) * E_PAR;
                piece(rd, eb, lane < 32 ? 16u * lane : 0x80000000u);
                piece(rs, eb + 1024, lane >= 32 ? 16u * (unsigned)(lane - 32) : 0x80000000u);
                if (PROJ) {
                    const __amdgpu_buffer_rsrc_t rd2 = make_rsrc_sgpr(p.scale2 + e_n0, (long long)(p.Cout - e_n0) * 4),
                                                 rs2 = make_rsrc_sgpr(p.shift2 + e_n0, (long long)(p.Cout - e_n0) * 4);
                    piece(rd2, eb + 2048, lane < 32 ? 16u * lane : 0x80000000u);
                    piece(rs2, eb + 3072, lane >= 32 ? 16u * (unsigned)(lane - 32) : 0x80000000u);
                }
            }
            issue_a();                                          // step g + 2: its slot held step g - 1, released at the last barrier
            issue_b();                                          // step g + 3: likewise
            W4_STAMP(0);
            // everything older than this iteration's pieces has landed (resident weights: from its second iteration on a loader issues none)
            if (p.b_resident && g + 3u >= (unsigned)NB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APW + BPW) : "memory");
            W4_STAMP(1);
            step_barrier();                                     // B_g: activations of step g + 1 and weights of step g + 2 are there
            W4_STAMP(2);
            if (++ckt == (unsigned)KT) { ckt = 0; ++ci; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W4_STAMP_FLUSH;
        return;
    }

    // =================================== MFMA waves 0..3 ===================================
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l16 = lane & 15, lq = lane >> 4;
    // fragment addresses: row (16 (wm RB + rb) + l16), chunk (4 half + lq) ^ (l16 & 6); the row block is an immediate
    const unsigned a0 = (unsigned)((wm * RB * 16 + l16) * ROWB + 16 * (lq ^ (l16 & 6)));
    const unsigned b0 = (unsigned)(B_OFF + (wn * 64 + l16) * ROWB + 16 * (lq ^ (l16 & 6)));

    f32x4 acc[RB][4];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    unsigned m0;
    int n0;
    unsigned ci = 0;
    int ckt = 0;
    tile_origin(0, m0, n0);

    bf16x8 bfr[4][2];        // weight fragments of the current step (re-loaded one by one for the next step in its last block)
    bf16x8 ar[3][2];         // activation fragments: ring over row blocks, two blocks ahead of the MFMAs
    auto lda = [&](const unsigned (&ab)[2], int rb, int half) __attribute__((always_inline)) -> bf16x8 {
        return *(const bf16x8*)(smem + ab[half] + rb * 16 * ROWB);
    };
    auto ldb = [&](const unsigned (&bb)[2], int nb, int half) __attribute__((always_inline)) -> bf16x8 {
        return *(const bf16x8*)(smem + bb[half] + nb * 16 * ROWB);
    };
    // epilogue geometry: lane (l16, lq) holds, per row block, pixel l16 x channels 64 wn + 32 j + 8 lq .. + 7 (j = 0, 1)
    const unsigned ylane = ((unsigned)(wm * RB * 16 + l16) * (unsigned)p.Cout + (unsigned)(wn * 64 + 8 * lq)) * 2u;
    f32x4 rres[RB][2];

    step_barrier();                                         // P
    {
        const unsigned bfirst[2] = {b0, b0 ^ 64u};
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { bfr[nb][0] = ldb(bfirst, nb, 0); bfr[nb][1] = ldb(bfirst, nb, 1); }
    }
    W4_STAMP_DECL;

    for (unsigned g = 0; g < nsteps; ++g) {
        const unsigned as0 = a0 + (g % NA) * ASTAGE;
        const unsigned acur[2] = {as0, as0 ^ 64u};
        const unsigned bn0 = b0 + ((g + 1u) & (NB - 1)) * BSTAGE;
        const unsigned bnext[2] = {bn0, bn0 ^ 64u};
        const bool last = ckt == KT - 1;
        if (!PROJ && last && p.res) {
            // the tile's residual fragments, requested now: there when the epilogue starts (rows past M: out-of-range offset, zeros)
            const long long yorg = ((long long)m0 * p.Cout + n0) * 2ll, ybytes = ((long long)(p.M - m0) * p.Cout - n0) * 2ll;
            const __amdgpu_buffer_rsrc_t rr = make_rsrc_sgpr((const char*)p.res + yorg, ybytes);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    rres[rb][j] = __builtin_bit_cast(hsefr_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ylane + 64u * (unsigned)j, __builtin_amdgcn_readfirstlane((unsigned)(16 * rb) * (unsigned)p.Cout * 2u), W4_RESAUX));
        }
#pragma unroll
        for (int blk = 0; blk < 2 && blk < RB; ++blk) { ar[blk][0] = lda(acur, blk, 0); ar[blk][1] = lda(acur, blk, 1); }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (rb + 2 < RB) { ar[(rb + 2) % 3][0] = lda(acur, rb + 2, 0); ar[(rb + 2) % 3][1] = lda(acur, rb + 2, 1); }
            const bf16x8 x0 = ar[rb % 3][0], x1 = ar[rb % 3][1];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nb][0], x0, acc[rb][nb], 0, 0, 0);
                if (rb == RB - 1) bfr[nb][0] = ldb(bnext, nb, 0);       // last use: the next step's fragment takes its place
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nb][1], x1, acc[rb][nb], 0, 0, 0);
                if (rb == RB - 1) bfr[nb][1] = ldb(bnext, nb, 1);
            }
            if (rb == RB - 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            } else if (rb + 2 < RB) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
        }
        W4_STAMP(0);
        step_barrier();                                     // B_g
        W4_STAMP(1);
        __builtin_amdgcn_sched_barrier(0);
        ++ckt;
        if (PROJ && ckt == KT2) {
            // the projection is complete: scale2 / shift2, rounded to bf16 where its tensor used to be stored, parked as bf16 pairs in
            // the registers (and the layout) of the plain kernel's residual fragments; the accumulators start again
            f32x4 e_sc[4], e_sh[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int cch = wn * 64 + 32 * (v >> 1) + 8 * lq + 4 * (v & 1);
                e_sc[v] = *(const f32x4*)(smem + E_OFF + (ci & 1u) * E_PAR + 2048 + cch * 4);
                e_sh[v] = *(const f32x4*)(smem + E_OFF + (ci & 1u) * E_PAR + 3072 + 512 + cch * 4);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * h + e] = fmaf(acc[rb][2 * j + h][e], e_sc[2 * j + h][e], e_sh[2 * j + h][e]);
#pragma unroll
                    for (int d = 0; d < 4; ++d) rres[rb][j][d] = __uint_as_float(hsefr_pack_bf16x2(v[2 * d], v[2 * d + 1]));
                }
            zero_acc();
        }
        if (ckt == KT) {
            ckt = 0;
            const long long yorg = ((long long)m0 * p.Cout + n0) * 2ll, ybytes = ((long long)(p.M - m0) * p.Cout - n0) * 2ll;
            const __amdgpu_buffer_rsrc_t ry = make_rsrc_sgpr((char*)p.y + yorg, ybytes);
            f32x4 e_sc[4], e_sh[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int cch = wn * 64 + 32 * (v >> 1) + 8 * lq + 4 * (v & 1);
                e_sc[v] = *(const f32x4*)(smem + E_OFF + (ci & 1u) * E_PAR + cch * 4);
                e_sh[v] = *(const f32x4*)(smem + E_OFF + (ci & 1u) * E_PAR + 1024 + 512 + cch * 4);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(16 * rb) * (unsigned)p.Cout * 2u);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * h + e] = fmaf(acc[rb][2 * j + h][e], e_sc[2 * j + h][e], e_sh[2 * j + h][e]);
                    if (PROJ || p.res) {
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const unsigned rw2 = __float_as_uint(rres[rb][j][d]);
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
                    // (rows past M belong to no tile: the resource ends with the tensor and the hardware drops them)
                    bstore16_welded(o, ry, ylane + 64u * (unsigned)j, soff);
                }
            }
            zero_acc();
            tile_origin(++ci, m0, n0);
            W4_STAMP(2);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // stores issued from asm: drained before the wave ends
    W4_STAMP_FLUSH;
}


HSEFR_KNOB(g_w4_off, 0);    // dev builds: 1 = never use this kernel, 2 = for every shape it covers
HSEFR_KNOB(g_w4_bres, 1);   // dev builds: 0 = reload the weight stages every step also where they could stay resident (A/B timing)

template <int RB, int WAVES_M, bool PROJ = false>
int launch_w4(W4Params& p, hipStream_t s) {
    constexpr int BM = WAVES_M * RB * 16, BN = (4 / WAVES_M) * 64;
    const long long tiles_m = ((long long)p.M + BM - 1) / BM;
    p.tiles_n = (unsigned)(p.Cout / BN);
    const long long total = tiles_m * p.tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv1x1_w4: too many tiles");
    p.total_tiles = (unsigned)total;
    const unsigned g = (unsigned)(total < 256 ? total : 256);
    // resident weights: the steps of a tile divide the ring (K = 64 / 128 / 256) and the grid's stride keeps a workgroup on one channel origin
    // (xcd_remap_dir permutes inside blocks of eight: conv1x1_bf16.hip's rule)
    p.b_resident = (!PROJ && g_w4_bres && total > g && (p.K == 64 || p.K == 128 || p.K == 256) && g % 8 == 0 && (g / 8) % p.tiles_n == 0) ? 1 : 0;
    HSEFR_LAUNCH((conv1x1_w4_bf16_kernel<RB, WAVES_M, PROJ>), dim3(g), dim3(512), 0, s, p);
    return launch_status("conv1x1_w4_bf16");
}

}  // namespace

#ifdef HSEFR_DEV
int read_w4_stamps(void* host_out, size_t bytes) {
#ifdef HSEFR_CD_STAMPS
    HSEFR_REQUIRE(bytes <= sizeof(unsigned long long) * 256 * 8 * 8, HSEFR_ERR_INVALID, "read_w4_stamps: too many bytes");
    HSEFR_HIP_CHECK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w4_stamps), bytes));
    return HSEFR_OK;
#else
    (void)host_out; (void)bytes;
    set_error("read_w4_stamps: library built without -DHSEFR_CD_STAMPS");
    return HSEFR_ERR_UNSUPPORTED;
#endif
}
void set_w4_off(int v) { g_w4_off = v; }
void set_w4_bres(int v) { g_w4_bres = v; }
#endif

bool conv1x1_w4_forced() { return g_w4_off == 2; }

bool conv1x1_w4_bf16_supported(long long n, int h, int w, int c, int oh, int ow, int cout, int stride) {
    if (g_w4_off == 1) return false;
    return c > 0 && c % 64 == 0 && cout > 0 && cout % 64 == 0 && n > 0 && stride >= 1 && oh > 0 && ow > 0 && (oh - 1) * stride < h && (ow - 1) * stride < w &&
           n * h * w * (long long)c * 2 < (1ll << 31) && n * oh * ow * (long long)cout * 2 < (1ll << 31) && n * oh * ow < (1ll << 31) &&
           (long long)cout * c * 2 < (1ll << 31);
}

// the shapes it measured faster on than the kernels before it (tools/kbench_conv.py, ResNet-50 at batch 128): the K-deep reductions of
// the 28- and 14-pixel stages, stride 1 and 2 (23.6 -> 20.7, 29.8 -> 27.5, 23.8 -> 20.2, 18.0 -> 16.2 us).  The increase layers are bound
// by what the CU's vector-memory path moves (operand fill + residual + stores: ~30 B/clk) on every kernel tried -- they stay on
// conv1x1_bf16.hip, whose two workgroups per CU overlap a tile's stores with the other's loads.
bool conv1x1_w4_bf16_preferred(long long pixels, int c, int cout, bool has_res) {
    // (+ the K = 512 increase layers of the 7-pixel stage, residual and all: 31.1 -> 27.9 us in the network at batch 128)
    if (has_res) return c >= 512 && cout % 128 == 0 && ((pixels + 223) / 224) * (cout / 128) >= 192;
    return c >= 256 && c <= 1024 && cout % 128 == 0 && pixels >= 20000;
}

int launch_conv1x1_w4_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y, int n, int h,
                           int w, int c, int oh, int ow, int cout, int stride, int act, hipStream_t s) {
    HSEFR_REQUIRE(conv1x1_w4_bf16_supported(n, h, w, c, oh, ow, cout, stride), HSEFR_ERR_UNSUPPORTED, "conv1x1_w4_bf16: shape not covered");
    HSEFR_REQUIRE(act == HSEFR_ACT_NONE || act == HSEFR_ACT_RELU || act == HSEFR_ACT_RELU6, HSEFR_ERR_UNSUPPORTED, "conv1x1_w4_bf16: act %d", act);
    W4Params p;
    p.x = x; p.wt = wt; p.scale = scale; p.shift = shift; p.res = res; p.y = y;
    p.x_bytes = (long long)n * h * w * c * 2;
    p.K = c; p.Cout = cout; p.stride = stride; p.H = h; p.W = w; p.OH = oh; p.OW = ow;
    p.act_lo = act == HSEFR_ACT_NONE ? -INFINITY : 0.f;
    p.act_hi = act == HSEFR_ACT_RELU6 ? 6.f : INFINITY;
    p.M = (unsigned)((long long)n * oh * ow);
    p.reverse = sweep_reverse();
    p.x2 = nullptr; p.wt2 = nullptr; p.scale2 = nullptr; p.shift2 = nullptr; p.x2_bytes = 0; p.K2 = 0; p.stride2 = 1; p.H2 = 0; p.W2 = 0;
    if (cout % 128 == 0) return launch_w4<7, 2>(p, s);
    return launch_w4<4, 4>(p, s);
}

// the increase layer of a stage's first block with its projected shortcut (PROJ): where the pair is matrix work -- the 28-, 14- and
// 7-pixel stages of ResNet-50 at batch 128, 39.5 GFLOP each -- this kernel's wide waves run it at the 3x3 layers' rate; the 56-pixel
// stage (K = K2 = 64: 308 MB for 26 GFLOP) stays on the register-staged kernel (measured: tools/kbench_conv.py)
bool conv1x1_w4_proj_preferred(long long pixels, int c, int c2, int cout) {
    if (g_w4_off == 1) return false;
    if (g_w4_off == 2) return cout % 128 == 0;
    // (at least three quarters of the CUs get a 224 x 128 tile: below that the register-staged kernel's 128 x 64 tiles fill the chip better)
    return cout % 128 == 0 && c + c2 >= 256 && ((pixels + 223) / 224) * (cout / 128) >= 192;
}

int launch_conv1x1_w4_proj_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* x2, const void* wt2,
                                const float* scale2, const float* shift2, void* y, int n, int oh, int ow, int c, int cout, int c2, int stride2,
                                int h2, int w2, int act, hipStream_t s) {
    HSEFR_REQUIRE(cout % 128 == 0 && conv1x1_w4_bf16_supported(n, oh, ow, c, oh, ow, cout, 1) && conv1x1_w4_bf16_supported(n, h2, w2, c2, oh, ow, cout, stride2),
                  HSEFR_ERR_UNSUPPORTED, "conv1x1_w4_proj_bf16: shape not covered");
    W4Params p;
    p.x = x; p.wt = wt; p.scale = scale; p.shift = shift; p.res = nullptr; p.y = y;
    p.x_bytes = (long long)n * oh * ow * c * 2;
    p.K = c; p.Cout = cout; p.stride = 1; p.H = oh; p.W = ow; p.OH = oh; p.OW = ow;
    p.act_lo = act == HSEFR_ACT_NONE ? -INFINITY : 0.f;
    p.act_hi = act == HSEFR_ACT_RELU6 ? 6.f : INFINITY;
    p.M = (unsigned)((long long)n * oh * ow);
    p.reverse = sweep_reverse();
    p.x2 = x2; p.wt2 = wt2; p.scale2 = scale2; p.shift2 = shift2;
    p.x2_bytes = (long long)n * h2 * w2 * c2 * 2;
    p.K2 = c2; p.stride2 = stride2; p.H2 = h2; p.W2 = w2;
    return launch_w4<7, 2, true>(p, s);
}

}  // namespace hsefr
