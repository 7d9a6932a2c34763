// General NHWC convolution in EXACT fp32 on the fp32 matrix pipe: the fp32-grade mode of ResNet-style graphs (the reference
// runs vgg2_resnet.pb in fp32: facerec_test.py:213, sess.run at :120) as an implicit GEMM on v_mfma_f32_32x32x2_f32.
//
//     y[p, n] = act( (sum_k x[pixel p, tap/channel k] * w[k, n]) * scale[n] + shift[n] + res[p, n] )
//     M = N*OH*OW pixels,  K = KH*KW*C (channel innermost, TF HWIO = [K][Cout] as it stands),  N = Cout
//
// the mode that meets the 1e-4 bar was a vector-FMA direct convolution at 1.2 k faces/s (0.06 of the fp32-MFMA
// peak).  v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain (bitwise a fmaf sequence), 64 cycles per SIMD for 4096 MACs: the pipe
// is so slow per byte that a plain design keeps it fed -- per wave and K pair 4 ds_read_b32 for 4 MFMAs (256 cycles).
//
//   * tile 128 pixels x BN channels (128 | 64), 4 waves as 2 x 2, wave tile 64 x BN/2 = 2 x BN/64 blocks of 32 x 32;
//   * K step of 16: the A rows are 64-byte runs of one tap (C % 16 == 0: every ResNet layer but the stem), gathered with two
//     16-byte loads per thread, rows on padding zero; C % 16 != 0 (the 7x7x3 stem) gathers value by value;
//   * both operands k-major in LDS ([16][BM + 4] / [16][BN + 4] floats): a fragment read is 32 consecutive floats of one k row
//     (conflict-free ds_read_b32), the next K step's global loads are in flight while the MFMAs of this one run (register
//     double buffering, one barrier pair per step);
//   * operands swapped (weights first), so a lane owns 4 consecutive output channels of a pixel: scale / shift / residual /
//     activation on float4s, 16-byte stores.
// Every output element is one fixed-order chain over K: results do not depend on the grid or the batch.
#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvF32Params {
    const float* x;      // [N,H,W,C]
    const float* w;      // [KH*KW*C][Cout]
    const float* scale;  // [Cout] or null
    const float* shift;  // [Cout] or null
    const float* res;    // [N,OH,OW,Cout] or null
    float* y;            // [N,OH,OW,Cout]
    int H, W, C, OH, OW, Cout, KH, KW, stride, pad_t, pad_l, act;
    long long M;         // N*OH*OW
    int K, KT;           // KH*KW*C, ceil(K / 16)
    unsigned tiles_n;
    int rs_stride, rs_h, rs_w;      // > 0: res is a LARGER map [N, rs_h, rs_w, Cout] read at every rs_stride-th pixel (lowering.subsample_stage_tails)
    hsefr_udiv d_ohow, d_ow;        // ... exact division by OH * OW and OW (common.h)
};

constexpr int BM = 128, BK = 16;

template <int BN, bool RUNS>
__global__ __launch_bounds__(256, 2) void conv_f32_mfma_kernel(ConvF32Params p) {
    constexpr int AP = BM + 4, BP = BN + 4;                 // k-row pitches in floats (16-byte aligned rows, banks shifted by 4 per k)
    constexpr int NB = BN / 64;                              // 32-column blocks per wave
    __shared__ __attribute__((aligned(16))) float As[BK * AP];
    __shared__ __attribute__((aligned(16))) float Bs[BK * BP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l32 = lane & 31, lh = lane >> 5;
    const unsigned tile = blockIdx.x;
    const unsigned tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const long long m0 = (long long)tm * BM;
    const int n0 = tn * BN;

    // ---- A gather geometry: thread = (row r = tid >> 2 (+ 64), k quad kq = tid & 3) -> floats 4 kq .. 4 kq + 3 of the step ----
    int a_n[2], a_oh[2], a_ow[2];
    bool a_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long long pix = m0 + (tid >> 2) + 64 * j;
        a_ok[j] = pix < p.M;
        const long long pc = a_ok[j] ? pix : 0;
        a_ow[j] = (int)(pc % p.OW);
        const long long t = pc / p.OW;
        a_oh[j] = (int)(t % p.OH);
        a_n[j] = (int)(t / p.OH);
    }
    const int kq = tid & 3;
    // ---- B geometry: thread = (k row = tid >> 4 (BN = 64) | tid >> 5 (+ 8), float4 column) ----
    constexpr int BQ = BN / 4;                               // float4 per B row
    constexpr int BROWS = 256 / BQ;                          // rows per pass: 16 | 8
    constexpr int BPASS = BK / BROWS;                        // 1 | 2
    const int b_col = (tid % BQ) * 4, b_row = tid / BQ;

    f32x4 ra[2], rb[BPASS];
    auto load_step = [&](int kt) {
        const int k0 = kt * BK;
        if constexpr (RUNS) {
            // the step lies inside one tap: k0 = (kh * KW + kw) * C + c0
            const int tap = k0 / p.C, c0 = k0 - tap * p.C;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ih = a_oh[j] * p.stride - p.pad_t + kh, iw = a_ow[j] * p.stride - p.pad_l + kw;
                const bool ok = a_ok[j] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                ra[j] = ok ? *(const f32x4*)(p.x + (((long long)a_n[j] * p.H + ih) * p.W + iw) * p.C + c0 + 4 * kq) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + 4 * kq + e;
                    const int tap = k / p.C, c = k - tap * p.C;
                    const int kh = tap / p.KW, kw = tap - kh * p.KW;
                    const int ih = a_oh[j] * p.stride - p.pad_t + kh, iw = a_ow[j] * p.stride - p.pad_l + kw;
                    const bool ok = a_ok[j] && k < p.K && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                    ra[j][e] = ok ? p.x[(((long long)a_n[j] * p.H + ih) * p.W + iw) * p.C + c] : 0.f;
                }
        }
#pragma unroll
        for (int q = 0; q < BPASS; ++q) {
            const int k = k0 + b_row + BROWS * q;
            rb[q] = k < p.K ? *(const f32x4*)(p.w + (long long)k * p.Cout + n0 + b_col) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) As[(4 * kq + e) * AP + (tid >> 2) + 64 * j] = ra[j][e];
#pragma unroll
        for (int q = 0; q < BPASS; ++q) *(f32x4*)(&Bs[(b_row + BROWS * q) * BP + b_col]) = rb[q];
    };

    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    load_step(0);
    for (int kt = 0; kt < p.KT; ++kt) {
        store_step();
        __syncthreads();
        if (kt + 1 < p.KT) load_step(kt + 1);               // in flight under this step's MFMAs
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            // lane (l32, lh): the weight fragment W[k = 2 kp + lh][channel], the activation fragment X[pixel][k = 2 kp + lh]
            float wf[NB], xf[2];
#pragma unroll
            for (int j = 0; j < NB; ++j) wf[j] = Bs[(2 * kp + lh) * BP + wn * (BN / 2) + 32 * j + l32];
#pragma unroll
            for (int i = 0; i < 2; ++i) xf[i] = As[(2 * kp + lh) * AP + wm * 64 + 32 * i + l32];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: lane (pixel l32 of block i, lh): acc[4 g + e] = channel 8 g + 4 lh + e of block j ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long pix = m0 + wm * 64 + 32 * i + l32;
        if (pix >= p.M) continue;
        long long rpix = pix;                 // the residual's pixel: the output's own, or (img, oy * s, ox * s) of the larger map
        if (p.rs_stride > 0) {
            const unsigned img = hsefr_udiv_do((unsigned)pix, p.d_ohow), rem = (unsigned)pix - img * (unsigned)(p.OH * p.OW);
            const unsigned oy = hsefr_udiv_do(rem, p.d_ow), ox = rem - oy * (unsigned)p.OW;
            rpix = ((long long)img * p.rs_h + oy * (unsigned)p.rs_stride) * p.rs_w + ox * (unsigned)p.rs_stride;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = n0 + wn * (BN / 2) + 32 * j + 8 * g + 4 * lh;
                const f32x4 sc = p.scale ? *(const f32x4*)(p.scale + ch) : (f32x4){1.f, 1.f, 1.f, 1.f};
                const f32x4 sh = p.shift ? *(const f32x4*)(p.shift + ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(acc[i][j][4 * g + e], sc[e], sh[e]);
                if (p.res) {
                    const f32x4 r = *(const f32x4*)(p.res + rpix * p.Cout + ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += r[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = apply_act_rt(o[e], p.act);
                *(f32x4*)(p.y + pix * p.Cout + ch) = o;
            }
    }
}

}  // namespace

bool conv_f32_mfma_supported(int c, int cout) { return cout % 64 == 0 && c > 0; }

int launch_conv_f32_mfma(const float* x, const float* w, const float* scale, const float* shift, const float* res, float* y, int n, int h,
                         int wd, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, int act, hipStream_t s,
                         int res_stride, int res_h, int res_w) {
    HSEFR_REQUIRE(n >= 0 && h > 0 && wd > 0 && c > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && oh > 0 && ow > 0, HSEFR_ERR_INVALID,
                  "conv_f32_mfma: bad shape");
    HSEFR_REQUIRE(cout % 64 == 0, HSEFR_ERR_UNSUPPORTED, "conv_f32_mfma: cout=%d must be a multiple of 64", cout);
    if (n == 0) return HSEFR_OK;
    ConvF32Params p;
    p.x = x; p.w = w; p.scale = scale; p.shift = shift; p.res = res; p.y = y;
    p.H = h; p.W = wd; p.C = c; p.OH = oh; p.OW = ow; p.Cout = cout; p.KH = kh; p.KW = kw; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
    p.act = act;
    p.M = (long long)n * oh * ow;
    p.rs_stride = 0; p.rs_h = p.rs_w = 0; p.d_ohow = p.d_ow = hsefr_udiv{0u, 0u};
    if (res_stride > 0) {
        HSEFR_REQUIRE(res && oh * ow > 1 && ow > 1 && (oh - 1) * res_stride < res_h && (ow - 1) * res_stride < res_w && p.M < (1ll << 32), HSEFR_ERR_INVALID,
                      "conv_f32_mfma: a %dx%d output is not a stride-%d view of a %dx%d residual", oh, ow, res_stride, res_h, res_w);
        p.rs_stride = res_stride; p.rs_h = res_h; p.rs_w = res_w;
        p.d_ohow = hsefr_udiv_make((unsigned)(oh * ow)); p.d_ow = hsefr_udiv_make((unsigned)ow);
    }
    p.K = kh * kw * c;
    p.KT = (p.K + BK - 1) / BK;
    const bool runs = c % 16 == 0;
    const int bn = cout % 128 == 0 ? 128 : 64;
    p.tiles_n = cout / bn;
    const long long tiles = ((p.M + BM - 1) / BM) * p.tiles_n;
    HSEFR_REQUIRE(tiles < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv_f32_mfma: grid too large");
#define HSEFR_CF32(BN_, R_) HSEFR_LAUNCH((conv_f32_mfma_kernel<BN_, R_>), dim3((unsigned)tiles), dim3(256), 0, s, p)
    if (bn == 128) { if (runs) HSEFR_CF32(128, true); else HSEFR_CF32(128, false); }
    else { if (runs) HSEFR_CF32(64, true); else HSEFR_CF32(64, false); }
#undef HSEFR_CF32
    return launch_status("conv_f32_mfma");
}

}  // namespace hsefr
