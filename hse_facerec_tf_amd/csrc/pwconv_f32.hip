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
// Tile: 128(M) x BN(N) x 32(K) per 256-thread workgroup, 4 waves as 2x2, each wave owning
// 64 x BN/2 as 32x32 MFMA blocks.  Both operands are K-contiguous in memory (X is NHWC, the
// weight is stored transposed [Cout][K]), so LDS tiles keep the global row layout and are
// filled by full-line float4 copies (8 lanes = one 128-B row segment).  The MFMA K index is
// a free permutation: lane (i, h) reads ONE float4 = k {8s+4h .. 8s+4h+3} of row i
// (ds_read_b128) and feeds element j to the j-th of four MFMAs; A and B use the same
// permutation, so each MFMA contracts k in {8s+j, 8s+4+j}.  LDS rows are padded to 36 floats
// (144 B): the 16 rows of every ds_read_b128 lane group then start on 16 distinct 4-bank
// slots -> conflict-free reads, and the float4 staging writes (8 lanes per row) are too.
// Global->LDS is register-staged and double-buffered: tile k+1 is loaded before the MFMAs of
// tile k and written after them (one barrier per K-tile).
// Workgroup ids are XCD-remapped so the N-tiles that re-read one X tile share an L2.
#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_ROW = 36;  // floats per LDS row (32 + 4 pad)

template <int BN, int ACT>
__global__ __launch_bounds__(256) void pwconv_f32_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                         const float* __restrict__ shift, float* __restrict__ y,
                                                         long long M, int K, int Cout, unsigned tiles_n,
                                                         unsigned nwg) {
    constexpr int WN = BN / 2;    // columns per wave
    constexpr int NT = WN / 32;   // 32-wide MFMA blocks per wave along N
    constexpr int BP = BN / 32;   // staging passes for the weight tile
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDS_ROW];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDS_ROW];

    const unsigned bid = xcd_remap(blockIdx.x, nwg);
    const unsigned tile_n = bid % tiles_n;
    const unsigned tile_m = bid / tiles_n;
    const long long m0 = (long long)tile_m * BM;
    const int n0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // staging: thread -> (row srow + 32p, float4 column skq)
    const int srow = tid >> 3, skq = tid & 7;
    const float* ag[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        long long r = m0 + srow + 32 * p;
        if (r > M - 1) r = M - 1;  // tail rows: read a valid row, never stored
        ag[p] = x + r * K + 4 * skq;
    }
    const float* bg = wt + (long long)(n0 + srow) * K + 4 * skq;

    f32x4 ra[4], rb[BP];
    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < 4; ++p) ra[p] = *(const f32x4*)(ag[p] + kt * BK);
#pragma unroll
        for (int p = 0; p < BP; ++p) rb[p] = *(const f32x4*)(bg + (long long)32 * p * K + kt * BK);
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) *(f32x4*)(&As[buf][(srow + 32 * p) * LDS_ROW + 4 * skq]) = ra[p];
#pragma unroll
        for (int p = 0; p < BP; ++p) *(f32x4*)(&Bs[buf][(srow + 32 * p) * LDS_ROW + 4 * skq]) = rb[p];
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int KT = K / BK;
    gload(0);
    swrite(0);
    __syncthreads();

    const int a_off = (wm * 64 + li) * LDS_ROW + 4 * lh;
    const int b_off = (wn * WN + li) * LDS_ROW + 4 * lh;

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) gload(kt + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 a[2], b[NT];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = *(const f32x4*)(&As[cur][a_off + mi * 32 * LDS_ROW + 8 * s]);
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) b[ni] = *(const f32x4*)(&Bs[cur][b_off + ni * 32 * LDS_ROW + 8 * s]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < KT) swrite(cur ^ 1);
        __syncthreads();
    }

    // Epilogue.  C/D map of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5).
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int col = n0 + wn * WN + ni * 32 + li;
        const float sh = shift[col];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const long long rbase = m0 + wm * 64 + mi * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < M) y[row * Cout + col] = apply_act<ACT>(acc[mi][ni][r] + sh);
            }
        }
    }
}

template <int BN>
int launch_bn(const float* x, const float* wt, const float* shift, float* y, long long m, int k, int cout,
              int act, hipStream_t s) {
    const long long tiles_m = (m + BM - 1) / BM;
    const unsigned tiles_n = cout / BN;
    const long long nwg = tiles_m * tiles_n;
    HSEFR_REQUIRE(nwg < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "pwconv: grid too large");
    dim3 grid((unsigned)nwg), block(256);
#define HSEFR_PW_LAUNCH(A) \
    hipLaunchKernelGGL((pwconv_f32_kernel<BN, A>), grid, block, 0, s, x, wt, shift, y, m, k, cout, tiles_n, (unsigned)nwg)
    if (act == HSEFR_ACT_RELU6) HSEFR_PW_LAUNCH(HSEFR_ACT_RELU6);
    else if (act == HSEFR_ACT_RELU) HSEFR_PW_LAUNCH(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_NONE) HSEFR_PW_LAUNCH(HSEFR_ACT_NONE);
    else { set_error("pwconv: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_PW_LAUNCH
    return launch_status("pwconv_f32");
}

}  // namespace

int launch_pwconv_f32(const float* x, const float* wgt_t, const float* shift, float* y, long long m, int k,
                      int cout, int act, hipStream_t s) {
    HSEFR_REQUIRE(k > 0 && k % BK == 0, HSEFR_ERR_UNSUPPORTED, "pwconv: k=%d must be a multiple of %d", k, BK);
    HSEFR_REQUIRE(cout > 0 && cout % 64 == 0, HSEFR_ERR_UNSUPPORTED, "pwconv: cout=%d must be a multiple of 64", cout);
    HSEFR_REQUIRE(m >= 0, HSEFR_ERR_INVALID, "pwconv: m=%lld", m);
    if (m == 0) return HSEFR_OK;
    if (cout % 128 == 0) return launch_bn<128>(x, wgt_t, shift, y, m, k, cout, act, s);
    return launch_bn<64>(x, wgt_t, shift, y, m, k, cout, act, s);
}

}  // namespace hsefr
