// Pointwise (1x1) convolution + shift + ReLU6 as an fp32-MFMA GEMM, NHWC fp32, gfx950.
//
// Replaces graph nodes Conv2D(k=[1,1,Cin,Cout], BN scale pre-folded) -> Add shift -> Relu ->
// Minimum 6 -> Maximum 0 (e.g. #45-49), run by tf_sess.run at facerec_test.py:120 /
// facial_analysis.py:109.  95 % of the trunk's FLOPs.
//
//   Y[m, n] = act( sum_k X[m, k] * Wt[n, k] + shift[n] ),  m = image*H*W + pixel (NHWC rows)
//
// Exact fp32: v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf chain (no TF32/xf32 on gfx950),
// at 64 FLOP/clk/SIMD = the 157 TF/s fp32-matrix peak that bounds layers pw_3..pw_13; pw_1/2
// (K = 32/64) are HBM-bound.
//
// Structure (256 threads = 4 waves as 2x2 over a BM x BN tile, BK = 32):
//  * Both operands are K-contiguous in memory (X is NHWC, the weight is stored transposed
//    [Cout][K]); LDS tiles keep that row layout (128-B rows) and are filled by full-line float4
//    copies -- 8 lanes = one 128-B row segment.
//  * The MFMA K index is a free permutation: lane (i, h) reads ONE float4 = k {8s+4h..8s+4h+3}
//    of row i (ds_read_b128) and feeds element j to the j-th of four MFMAs; A and B use the same
//    permutation, so MFMA j of chunk s contracts k in {8s+j, 8s+4+j}.
//  * LDS rows are unpadded; the 16-B chunk c of row r lives at chunk position c ^ ((r >> 1) & 7).
//    With 128-B rows two consecutive rows cover the 64 banks, and every ds_read_b128 lane group
//    (16 rows, one logical chunk) then hits 16 distinct (row parity, position) slots:
//    conflict-free reads; the staging writes (8 lanes = 8 chunks of one row) are too.
//  * Persistent workgroups walk tiles t = blockIdx.x, +gridDim.x, ...; the (tile, k-tile) steps
//    form ONE software pipeline: the global loads of the next step (also across a tile boundary)
//    are issued before the MFMAs of the current one and written to the other LDS buffer after
//    them (one barrier per step), so the epilogue stores and the next tile's first loads overlap.
//  * Tile ids are XCD-remapped: the N-tiles that re-read one X tile run on one XCD (shared L2).
//  * Tile shape is chosen per layer so that tiles divide evenly over 256 CUs x resident
//    workgroups (tile quantisation cost 25-44 % with one fixed 128x128 tile).
#include <stdlib.h>

#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;

__device__ __forceinline__ int swz(int row, int chunk) { return row * BK + 4 * (chunk ^ ((row >> 1) & 7)); }

template <int BM, int BN, int OCC, int ACT>
__global__ __launch_bounds__(256, OCC) void pwconv_f32_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                         const float* __restrict__ shift, float* __restrict__ y,
                                                         long long M, int K, int Cout, unsigned tiles_n,
                                                         unsigned total_tiles, int ablate) {
    constexpr int WM = BM / 2, WN = BN / 2;  // wave tile
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int AP = BM / 32, BP = BN / 32;  // staging passes (32 rows x 8 float4 per pass)
    __shared__ __attribute__((aligned(16))) float As[2][BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int srow = tid >> 3, skq = tid & 7;
    const int KT = K / BK;

    unsigned t = blockIdx.x;
    if (t >= total_tiles) return;

    // global pointers of the staging thread for a tile
    const float* ag[AP];
    const float* bg;
    long long m0;
    int n0;
    auto setup = [&](unsigned tile) {
        const unsigned lt = xcd_remap(tile, total_tiles);
        const unsigned tn = lt % tiles_n, tm = lt / tiles_n;
        m0 = (long long)tm * BM;
        n0 = tn * BN;
    };
    auto setup_ptrs = [&](long long mm0, int nn0) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            long long r = mm0 + srow + 32 * p;
            if (r > M - 1) r = M - 1;  // tail rows: read a valid row, never stored
            ag[p] = x + r * K + 4 * skq;
        }
        bg = wt + (long long)(nn0 + srow) * K + 4 * skq;
    };

    f32x4 ra[AP], rb[BP];
    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < AP; ++p) ra[p] = *(const f32x4*)(ag[p] + kt * BK);
#pragma unroll
        for (int p = 0; p < BP; ++p) rb[p] = *(const f32x4*)(bg + (long long)32 * p * K + kt * BK);
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int p = 0; p < AP; ++p) *(f32x4*)(&As[buf][swz(srow + 32 * p, skq)]) = ra[p];
#pragma unroll
        for (int p = 0; p < BP; ++p) *(f32x4*)(&Bs[buf][swz(srow + 32 * p, skq)]) = rb[p];
    };

    f32x16 acc[MI][NI];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    };
    zero_acc();

    setup(t);
    setup_ptrs(m0, n0);
    gload(0);
    swrite(0);
    __syncthreads();
    int buf = 0;

    const int arow = wm * WM + li, brow = wn * WN + li;

    while (true) {
        const unsigned tnext = t + gridDim.x;
        const bool more_tiles = tnext < total_tiles;
        for (int kt = 0; kt < KT; ++kt) {
            const bool last = kt + 1 == KT;
            const bool has_next = !last || more_tiles;
            long long m0n = m0;
            int n0n = n0;
            if (has_next && !(ablate & 1)) {
                if (last) {
                    const unsigned lt = xcd_remap(tnext, total_tiles);
                    m0n = (long long)(lt / tiles_n) * BM;
                    n0n = (lt % tiles_n) * BN;
                    setup_ptrs(m0n, n0n);
                    gload(0);
                } else {
                    gload(kt + 1);
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x4 a[MI], b[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[mi] = *(const f32x4*)(&As[buf][swz(arow + mi * 32, 2 * s + lh)]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[ni] = *(const f32x4*)(&Bs[buf][swz(brow + ni * 32, 2 * s + lh)]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
            }
            if (has_next && !(ablate & 1)) swrite(buf ^ 1);
            __syncthreads();
            buf ^= 1;
            if (last) {
                // Epilogue of tile (m0, n0).  C/D map of the 32x32 MFMA: column = lane & 31,
                // row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5).  Full tiles store unconditionally:
                // a per-store bounds branch makes hipcc put s_waitcnt vmcnt(0) in front of EVERY
                // store (each one then waits for the previous to retire).
                const bool full_tile = m0 + BM <= M;
                if (ablate & 2) {   // timing-only ablation: keep the accumulators live, skip the stores
                    float live = 0.f;
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                            for (int r = 0; r < 16; ++r) live += acc[mi][ni][r];
                    if (live == 1.2345e-30f) y[0] = live;
                } else
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int col = n0 + wn * WN + ni * 32 + li;
                    const float sh = shift[col];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const long long rbase = m0 + wm * WM + mi * 32 + 4 * lh;
                        float* yp = y + rbase * Cout + col;
                        if (full_tile) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                yp[(long long)((r & 3) + 8 * (r >> 2)) * Cout] = apply_act<ACT>(acc[mi][ni][r] + sh);
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int dr = (r & 3) + 8 * (r >> 2);
                                if (rbase + dr < M) yp[(long long)dr * Cout] = apply_act<ACT>(acc[mi][ni][r] + sh);
                            }
                        }
                    }
                }
                zero_acc();
                m0 = m0n;
                n0 = n0n;
            }
        }
        if (!more_tiles) break;
        t = tnext;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant: the tiles go global -> LDS with global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass).
// One wave-instruction deposits 64 x 16 B = 8 consecutive 128-B LDS rows; the XOR swizzle therefore moves to the
// SOURCE side: lane (row r, position p) fetches logical chunk p ^ ((r >> 1) & 7) of its row (cdna guide rule 21:
// linear destination + swizzled source + the same swizzle on the ds_read).  All LDS lives in ONE array (a second
// __shared__ object next to an LDS-DMA target makes hipcc drain vmcnt before every ds_read).  Measured against the
// register-staged kernel above with tools/kbench.py; selectable with hsefr_debug_set("pw_dma", 0|1).
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int OCC, int ACT>
__global__ __launch_bounds__(256, OCC) void pwconv_f32_dma_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                  const float* __restrict__ shift, float* __restrict__ y,
                                                                  long long M, int K, int Cout, unsigned tiles_n,
                                                                  unsigned total_tiles) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int AI = BM / 32, BI = BN / 32;        // DMA wave-instructions per wave per K-tile (8 rows each)
    constexpr int STAGE = (BM + BN) * BK;            // floats per buffer: A rows then B rows
    __shared__ __attribute__((aligned(1024))) float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int KT = K / BK;

    unsigned t = blockIdx.x;
    if (t >= total_tiles) return;

    const float* asrc[AI];
    const float* bsrc[BI];
    long long m0;
    int n0;
    auto setup_ptrs = [&](long long mm0, int nn0) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int r = (wave * AI + i) * 8 + (lane >> 3);
            long long gr = mm0 + r;
            if (gr > M - 1) gr = M - 1;            // tail rows: read a valid row, never stored
            asrc[i] = x + gr * K + 4 * ((lane & 7) ^ ((r >> 1) & 7));
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int r = (wave * BI + i) * 8 + (lane >> 3);
            bsrc[i] = wt + (long long)(nn0 + r) * K + 4 * ((lane & 7) ^ ((r >> 1) & 7));
        }
    };
    auto dma = [&](int kt, int buf) {
        float* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < AI; ++i) glds16(asrc[i] + kt * BK, base + (wave * AI + i) * 8 * BK);
#pragma unroll
        for (int i = 0; i < BI; ++i) glds16(bsrc[i] + kt * BK, base + BM * BK + (wave * BI + i) * 8 * BK);
    };

    f32x16 acc[MI][NI];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    };
    zero_acc();
    {
        const unsigned lt = xcd_remap(t, total_tiles);
        m0 = (long long)(lt / tiles_n) * BM;
        n0 = (lt % tiles_n) * BN;
    }
    setup_ptrs(m0, n0);
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    const int arow = wm * WM + li, brow = wn * WN + li;

    while (true) {
        const unsigned tnext = t + gridDim.x;
        const bool more_tiles = tnext < total_tiles;
        for (int kt = 0; kt < KT; ++kt) {
            const bool last = kt + 1 == KT;
            const bool has_next = !last || more_tiles;
            long long m0n = m0;
            int n0n = n0;
            if (has_next) {
                if (last) {
                    const unsigned lt = xcd_remap(tnext, total_tiles);
                    m0n = (long long)(lt / tiles_n) * BM;
                    n0n = (lt % tiles_n) * BN;
                    setup_ptrs(m0n, n0n);
                    dma(0, buf ^ 1);
                } else {
                    dma(kt + 1, buf ^ 1);
                }
            }
            const float* As = smem + buf * STAGE;
            const float* Bs = As + BM * BK;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x4 a[MI], b[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[mi] = *(const f32x4*)(&As[swz(arow + mi * 32, 2 * s + lh)]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[ni] = *(const f32x4*)(&Bs[swz(brow + ni * 32, 2 * s + lh)]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the next tile have landed
            __syncthreads();
            buf ^= 1;
            if (last) {
                const bool full_tile = m0 + BM <= M;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int col = n0 + wn * WN + ni * 32 + li;
                    const float sh = shift[col];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const long long rbase = m0 + wm * WM + mi * 32 + 4 * lh;
                        float* yp = y + rbase * Cout + col;
                        if (full_tile) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                yp[(long long)((r & 3) + 8 * (r >> 2)) * Cout] = apply_act<ACT>(acc[mi][ni][r] + sh);
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int dr = (r & 3) + 8 * (r >> 2);
                                if (rbase + dr < M) yp[(long long)dr * Cout] = apply_act<ACT>(acc[mi][ni][r] + sh);
                            }
                        }
                    }
                }
                zero_acc();
                m0 = m0n;
                n0 = n0n;
            }
        }
        if (!more_tiles) break;
        t = tnext;
    }
}

struct TileCfg { int bm, bn, occ; };

// Work per CU if tiles are dealt evenly: ceil(T / 256) tiles of bm*bn; relative tile efficiency
// favours big tiles (less L2->LDS traffic per MFMA).
TileCfg choose_tile(long long m, int cout, int forced) {
    const TileCfg cands[3] = {{128, 128, 2}, {128, 64, 3}, {64, 64, 3}};
    const double eff[3] = {1.00, 0.97, 0.88};
    if (forced >= 0 && forced < 3 && cout % cands[forced].bn == 0) return cands[forced];
    int best = -1;
    double best_cost = 0;
    for (int i = 0; i < 3; ++i) {
        if (cout % cands[i].bn) continue;
        const long long tiles = ((m + cands[i].bm - 1) / cands[i].bm) * (cout / cands[i].bn);
        const long long slots = 256ll * cands[i].occ;
        const long long rounds = (tiles + slots - 1) / slots;
        // time ~ rounds * (work of `occ` co-resident tiles on one CU)
        const double cost = (double)rounds * cands[i].occ * cands[i].bm * cands[i].bn / eff[i];
        if (best < 0 || cost < best_cost) { best = i; best_cost = cost; }
    }
    return cands[best];
}

HSEFR_KNOB(g_pw_dma, 1);       // 1 = LDS-DMA staging (default), 0 = register staging
HSEFR_KNOB(g_pw_ablate, 0);    // timing-only ablations (results WRONG): 1 = no global loads after the first tile, 2 = no stores
HSEFR_KNOB(g_forced_tile, -1);  // dev builds: 0 = 128x128, 1 = 128x64, 2 = 64x64

template <int BM, int BN, int OCC>
int launch_cfg(const float* x, const float* wt, const float* shift, float* y, long long m, int k, int cout,
               int act, hipStream_t s) {
    constexpr int occ = OCC;
    const long long tiles_m = (m + BM - 1) / BM;
    const unsigned tiles_n = cout / BN;
    const long long total = tiles_m * tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "pwconv: too many tiles");
    const long long g = total < 256ll * occ ? total : 256ll * occ;
    dim3 grid((unsigned)g), block(256);
#define HSEFR_PW_LAUNCH(A)                                                                                          \
    do {                                                                                                            \
        if (g_pw_dma && !g_pw_ablate)                                                                               \
            HSEFR_LAUNCH((pwconv_f32_dma_kernel<BM, BN, OCC, A>), grid, block, 0, s, x, wt, shift, y, m, k, cout, \
                               tiles_n, (unsigned)total);                                                            \
        else                                                                                                        \
            HSEFR_LAUNCH((pwconv_f32_kernel<BM, BN, OCC, A>), grid, block, 0, s, x, wt, shift, y, m, k, cout,  \
                               tiles_n, (unsigned)total, g_pw_ablate);                                               \
    } while (0)
    if (act == HSEFR_ACT_RELU6) HSEFR_PW_LAUNCH(HSEFR_ACT_RELU6);
    else if (act == HSEFR_ACT_RELU) HSEFR_PW_LAUNCH(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_NONE) HSEFR_PW_LAUNCH(HSEFR_ACT_NONE);
    else { set_error("pwconv: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_PW_LAUNCH
    return launch_status("pwconv_f32");
}

}  // namespace

#ifdef HSEFR_DEV
void set_pw_tile(int v) { g_forced_tile = v; }
void set_pw_ablate(int v) { g_pw_ablate = v; }
void set_pw_dma(int v) { g_pw_dma = v; }
#endif

int launch_pwconv_f32(const float* x, const float* wgt_t, const float* shift, float* y, long long m, int k,
                      int cout, int act, hipStream_t s) {
    HSEFR_REQUIRE(k > 0 && k % BK == 0, HSEFR_ERR_UNSUPPORTED, "pwconv: k=%d must be a multiple of %d", k, BK);
    HSEFR_REQUIRE(cout > 0 && cout % 64 == 0, HSEFR_ERR_UNSUPPORTED, "pwconv: cout=%d must be a multiple of 64", cout);
    HSEFR_REQUIRE(m >= 0, HSEFR_ERR_INVALID, "pwconv: m=%lld", m);
    if (m == 0) return HSEFR_OK;
    const TileCfg c = choose_tile(m, cout, g_forced_tile);
    if (c.bm == 128 && c.bn == 128) return launch_cfg<128, 128, 2>(x, wgt_t, shift, y, m, k, cout, act, s);
    if (c.bm == 128 && c.bn == 64) return launch_cfg<128, 64, 3>(x, wgt_t, shift, y, m, k, cout, act, s);
    return launch_cfg<64, 64, 3>(x, wgt_t, shift, y, m, k, cout, act, s);
}

}  // namespace hsefr
