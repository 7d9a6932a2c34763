// Depthwise 3x3 convolution + per-channel scale + shift + ReLU6, NHWC fp32, gfx950.
//
// Replaces graph nodes DepthwiseConv2dNative -> Mul -> Add -> Relu -> Minimum -> Maximum
// (e.g. #35-39,#44 of the reference's frozen MobileNet; executed by tf_sess.run at
// facerec_test.py:120 / facial_analysis.py:109).
//
// HBM-bound (0.9-2.2 flop/B): the design goal is that every input byte leaves HBM once
// and every load/store instruction moves 1 KiB per wave.
//   * a thread owns ONE float4 of channels at ONE output column and walks DOWN a strip of
//     TH output rows, keeping the 3x3 input window in registers (sliding window: stride 1
//     loads 1 new input row per output row, stride 2 loads 2);
//   * consecutive lanes = consecutive (column, channel-group) float4s, which are contiguous
//     in NHWC, so each of the 3 horizontal taps is one fully coalesced 16 B/lane load; the
//     3x horizontal re-read is served by the CU's L1, not HBM;
//   * vertically adjacent strips (which share 2 halo rows) get workgroup ids on the same
//     XCD (xcd_remap) so the halo re-read hits that XCD's L2.
// Algorithmic bytes per image: 4*C*(H*W + OH*OW) + 36*C + 8*C   (SURVEY 8d).
#include "common.h"

namespace hsefr {

namespace {

// knobs (constants in the product build; see HSEFR_KNOB in common.h)
HSEFR_KNOB(g_dw_th, 0);         // forced strip height
HSEFR_KNOB(g_dw_variant, 0);    // load-policy / grid variants
HSEFR_KNOB(g_dw_look, 4);       // 2..5 rows of load lookahead, stride-1 kernel (measured in situ: 4 is best)
HSEFR_KNOB(g_dw_look2, 2);      // 2 | 4 = one | two iterations of lookahead, stride-2 kernel

typedef float f4 __attribute__((ext_vector_type(4)));

struct DwParams {
    const float4* x;
    const float4* w;      // [9][C4] float4 (TF [3,3,C,1] viewed as float4 over C)
    const float4* scale;  // [C4]
    const float4* shift;  // [C4]
    float4* y;
    int H, W, C4, OH, OW, pad_t, pad_l, TH, tiles_h, tiles_x;
    unsigned nwg;
    int reverse;  // sweep direction (common.h)
    int variant;  // 0 = real kernel; timing-only ablations: 1 = one load per row, 2 = no stores
    float a_scale;  // SPLIT kernels: 2^a_log2
};

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ntload(const float4* p) {
    const f4 v = __builtin_nontemporal_load((const f4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void ntstore(float4 v, float4* p) {
    f4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, (f4*)p);
}

__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// LOOK = rows requested ahead of the one being consumed (stride 1): 2 at 4 workgroups per CU, or 3 at 3 (more unique bytes
// in flight per CU: the waves of this kernel spend two thirds of their life waiting on memory).
// SPLIT = 1: the result is stored PRE-SPLIT for the split-f16 GEMM that consumes it (pwconv_ps.hip): per pixel and
// 32-channel group one 128-byte "split row" [hi(32 x f16) | lo(32 x f16)], hi = f16(v * 2^a_log2), lo = f16(v * 2^a_log2 - hi)
// -- the very operations pwconv_f16s.hip applies on its way into LDS, so the GEMM sees the same bits either way.  The
// tensor keeps its size (4 B per element).  A thread owns 4 channels = 8 B of the hi half and 8 B of the lo half; lane
// pairs (the quads 2j, 2j+1 of a group) swap one 8-byte half by DPP, so that every lane still issues ONE 16-byte store
// and a wave still writes whole 128-byte lines: even quad -> [hi(2j) | hi(2j+1)], odd quad -> [lo(2j) | lo(2j+1)].
template <int STRIDE, int ACT, int NT, int LOOK, int SPLIT = 0>
__global__ __launch_bounds__(256, LOOK == 2 ? 4 : (LOOK <= 4 ? 3 : 2)) void dwconv3x3_kernel(DwParams p) {
    const unsigned bid = xcd_remap_dir(blockIdx.x, p.nwg, p.reverse);
    const int tx = bid % p.tiles_x;
    const int th = (bid / p.tiles_x) % p.tiles_h;
    const int n = bid / (p.tiles_x * p.tiles_h);
    const int t = tx * 256 + threadIdx.x;
    if (t >= p.OW * p.C4) return;
    const int ow = t / p.C4;
    const int c4 = t - ow * p.C4;

    // Branch-free borders: every load uses a clamped (always valid) address, and the padding is
    // applied arithmetically -- out-of-image COLUMNS by zeroing this thread's tap weights once
    // (they are constant down the strip), out-of-image ROWS by a 0/1 factor on that row's partial
    // sum.  No control flow in the row loop, so the compiler keeps counted s_waitcnt vmcnt(N) and
    // the row prefetches really stay in flight (a branch per load made it emit vmcnt(0) each time).
    const int iw0 = ow * STRIDE - p.pad_l;  // column of the left tap
    const float ml = iw0 >= 0 ? 1.f : 0.f, mm = (iw0 + 1 >= 0 && iw0 + 1 < p.W) ? 1.f : 0.f,
                mr = iw0 + 2 < p.W ? 1.f : 0.f;
    float4 wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float4 w = p.w[i * p.C4 + c4];
        const float m = (i % 3 == 0) ? ml : (i % 3 == 1 ? mm : mr);
        wk[i] = make_float4(w.x * m, w.y * m, w.z * m, w.w * m);
    }
    const float4 sc = p.scale[c4];
    const float4 sh = p.shift[c4];
    const int cl = max(iw0, 0) * p.C4, cm = min(max(iw0 + 1, 0), p.W - 1) * p.C4, cr = min(iw0 + 2, p.W - 1) * p.C4;
    const float4* xin = p.x + (size_t)n * p.H * p.W * p.C4 + c4;

    struct Row { float4 l, m, r; float k; };
    auto load_row = [&](int ih) {
        Row q;
        const int ihc = min(max(ih, 0), p.H - 1);
        const float4* row = xin + (size_t)ihc * p.W * p.C4;
        if (NT & 1) {
            q.l = ntload(row + cl);
            q.m = ntload(row + cm);
            q.r = ntload(row + cr);
        } else {
            q.l = row[cl];
            q.m = row[cm];
            q.r = row[cr];
        }
        q.k = (ih >= 0 && ih < p.H) ? 1.f : 0.f;
        return q;
    };
    auto row_sum = [&](const Row& q, int base) {
        float4 a = make_float4(q.l.x * wk[base].x, q.l.y * wk[base].y, q.l.z * wk[base].z, q.l.w * wk[base].w);
        a = fma4(q.m, wk[base + 1], a);
        a = fma4(q.r, wk[base + 2], a);
        return a;
    };

    const int oh0 = th * p.TH;
    const int oh1 = min(oh0 + p.TH, p.OH);
    // 16-B unit of this lane's store inside the pixel: fp32 layout = its channel quad; split rows = quad q of group gq goes
    // to unit q/2 of the hi half (even q) or 4 + q/2, the lo half (odd q)
    const int qg = c4 & 7;
    const int c4s = SPLIT ? (c4 & ~7) + ((qg & 1) ? 4 + (qg >> 1) : (qg >> 1)) : c4;
    float4* yout = p.y + ((size_t)n * p.OH * p.OW + ow) * p.C4 + c4s;

    auto compute_store = [&](int oh, const Row& a, const Row& b, const Row& c) {
        const float4 sa = row_sum(a, 0), sb = row_sum(b, 3), sc2 = row_sum(c, 6);
        float4 acc = make_float4(sa.x * a.k, sa.y * a.k, sa.z * a.k, sa.w * a.k);
        acc = make_float4(fmaf(sb.x, b.k, acc.x), fmaf(sb.y, b.k, acc.y), fmaf(sb.z, b.k, acc.z), fmaf(sb.w, b.k, acc.w));
        acc = make_float4(fmaf(sc2.x, c.k, acc.x), fmaf(sc2.y, c.k, acc.y), fmaf(sc2.z, c.k, acc.z), fmaf(sc2.w, c.k, acc.w));
        float4 o = fma4(acc, sc, sh);
        o.x = apply_act<ACT>(o.x);
        o.y = apply_act<ACT>(o.y);
        o.z = apply_act<ACT>(o.z);
        o.w = apply_act<ACT>(o.w);
        if (SPLIT) {
            const f4 v = f4{o.x, o.y, o.z, o.w} * p.a_scale;
            const f16x4 hi = __builtin_convertvector(v, f16x4);
            const f16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f4), f16x4);
            const u32x2 hb = __builtin_bit_cast(u32x2, hi), lb = __builtin_bit_cast(u32x2, lo);
            const bool odd = qg & 1;
            const u32x2 send = odd ? hb : lb;          // the even quad gives away its lo half, the odd quad its hi half
            u32x2 recv;                                // quad_perm [1,0,3,2]: swap with the neighbouring lane
            recv.x = (unsigned)__builtin_amdgcn_mov_dpp((int)send.x, 0xB1, 0xF, 0xF, true);
            recv.y = (unsigned)__builtin_amdgcn_mov_dpp((int)send.y, 0xB1, 0xF, 0xF, true);
            const u32x4 out = odd ? u32x4{recv.x, recv.y, lb.x, lb.y} : u32x4{hb.x, hb.y, recv.x, recv.y};
            const f4 ov = __builtin_bit_cast(f4, out);
            if (NT & 2) ntstore(make_float4(ov.x, ov.y, ov.z, ov.w), yout + (size_t)oh * p.OW * p.C4);
            else yout[(size_t)oh * p.OW * p.C4] = make_float4(ov.x, ov.y, ov.z, ov.w);
            return;
        }
        if (NT & 2) ntstore(o, yout + (size_t)oh * p.OW * p.C4);
        else yout[(size_t)oh * p.OW * p.C4] = o;
    };

    if (STRIDE == 1) {
        // sliding window, rows requested TWO iterations before they are consumed
        const int ih = oh0 - p.pad_t;
        Row r[LOOK + 2];     // r[0..2] = the window; r[3..] requested ahead
#pragma unroll
        for (int i = 0; i < LOOK + 2; ++i) r[i] = load_row(ih + i);
        for (int oh = oh0; oh < oh1; ++oh) {
            const Row nx = load_row(oh - p.pad_t + LOOK + 2);
            compute_store(oh, r[0], r[1], r[2]);
#pragma unroll
            for (int i = 0; i < LOOK + 1; ++i) r[i] = r[i + 1];
            r[LOOK + 1] = nx;
        }
    } else {
        // stride 2: two new rows per output row, requested LOOK/2 iterations ahead
        constexpr int LA = LOOK / 2;
        const int ih = oh0 * 2 - p.pad_t;
        Row r0 = load_row(ih), r1 = load_row(ih + 1), r2 = load_row(ih + 2);
        Row f[2 * LA];
#pragma unroll
        for (int i = 0; i < 2 * LA - 2; ++i) f[i] = load_row(ih + 3 + i);
        for (int oh = oh0; oh < oh1; ++oh) {
            f[2 * LA - 2] = load_row(oh * 2 - p.pad_t + 2 * LA + 1);
            f[2 * LA - 1] = load_row(oh * 2 - p.pad_t + 2 * LA + 2);
            compute_store(oh, r0, r1, r2);
            r0 = r2; r1 = f[0]; r2 = f[1];
#pragma unroll
            for (int i = 0; i < 2 * LA - 2; ++i) f[i] = f[i + 2];
        }
    }
}

}  // namespace

#ifdef HSEFR_DEV
void set_dw_th(int v) { g_dw_th = v; }
void set_dw_variant(int v) { g_dw_variant = v; }
void set_dw_look(int v) { g_dw_look = (v >= 2 && v <= 5) ? v : 2; }
void set_dw_look2(int v) { g_dw_look2 = v == 4 ? 4 : 2; }
#endif

static int launch_dwconv3x3_impl(const float* x, const float* wgt, const float* scale, const float* shift, float* y,
                                 int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh, int ow, int act,
                                 int a_log2, hipStream_t s) {
    HSEFR_REQUIRE(c % 4 == 0, HSEFR_ERR_UNSUPPORTED, "dwconv3x3: c=%d must be a multiple of 4", c);
    if (a_log2) {
        HSEFR_REQUIRE(c % 32 == 0, HSEFR_ERR_UNSUPPORTED, "dwconv3x3 (split output): c=%d must be a multiple of 32", c);
        HSEFR_REQUIRE(act == HSEFR_ACT_RELU6 && a_log2 > 0 && a_log2 <= 12, HSEFR_ERR_INVALID,
                      "dwconv3x3 (split output): needs the ReLU6 bound and a_log2 in [1, 12] (6 * 2^a_log2 < 32768), got act %d a_log2 %d",
                      act, a_log2);
    }
    HSEFR_REQUIRE(stride == 1 || stride == 2, HSEFR_ERR_UNSUPPORTED, "dwconv3x3: stride %d", stride);
    HSEFR_REQUIRE(n >= 0 && h > 0 && w > 0 && oh > 0 && ow > 0, HSEFR_ERR_INVALID, "dwconv3x3: bad shape");
    if (n == 0) return HSEFR_OK;
    DwParams p;
    p.x = (const float4*)x; p.w = (const float4*)wgt; p.scale = (const float4*)scale;
    p.shift = (const float4*)shift; p.y = (float4*)y;
    p.variant = g_dw_variant;
    p.reverse = sweep_reverse();
    p.a_scale = ldexpf(1.f, a_log2);
    p.H = h; p.W = w; p.C4 = c / 4; p.OH = oh; p.OW = ow; p.pad_t = pad_t; p.pad_l = pad_l;
    p.tiles_x = (ow * p.C4 + 255) / 256;
    // Strip height: tall strips amortise the 2-row halo, short ones balance the CUs.
    // measured in situ on MI355X (HSEFR_DEBUG=dw_th=N bench.py --layers): 24-row strips wherever the map has them
    // (with the deeper load lookahead a 24-row map is best walked whole: no halo rows re-read), 12 otherwise
    int th = oh >= 24 ? 24 : 12;
    if (th > oh) th = oh;
    while (th > 4 && (long long)n * p.tiles_x * ((oh + th - 1) / th) < 512) th = (th + 1) / 2;
    if (g_dw_th > 0) th = g_dw_th < oh ? g_dw_th : oh;  // tuning/debug only (hsefr_debug_set "dw_th")
    p.TH = th;
    p.tiles_h = (oh + th - 1) / th;
    const long long nwg = (long long)n * p.tiles_x * p.tiles_h;
    HSEFR_REQUIRE(nwg < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "dwconv3x3: grid too large");
    p.nwg = (unsigned)nwg;
    dim3 grid((unsigned)nwg), block(256);
#define HSEFR_DW_LAUNCH(S, A)                                                                      \
    do {                                                                                            \
        const int look = S == 1 ? g_dw_look : g_dw_look2;                                            \
        if (a_log2) {   /* split rows for the GEMM behind (ReLU6 only: checked above) */           \
            if (A == HSEFR_ACT_RELU6) {                                                             \
                if (S == 1) HSEFR_LAUNCH((dwconv3x3_kernel<S, HSEFR_ACT_RELU6, 0, 4, 1>), grid, block, 0, s, p); \
                else if (g_dw_variant & 4) HSEFR_LAUNCH((dwconv3x3_kernel<S, HSEFR_ACT_RELU6, 2, 2, 1>), grid, block, 0, s, p); \
                else HSEFR_LAUNCH((dwconv3x3_kernel<S, HSEFR_ACT_RELU6, 0, 2, 1>), grid, block, 0, s, p);      \
            }                                                                                       \
            break;                                                                                  \
        }                                                                                           \
        if (look == 3 && S == 1) { HSEFR_LAUNCH((dwconv3x3_kernel<S, A, 0, 3>), grid, block, 0, s, p); break; } \
        if (look == 4) { HSEFR_LAUNCH((dwconv3x3_kernel<S, A, 0, 4>), grid, block, 0, s, p); break; } \
        if (look == 5 && S == 1) { HSEFR_LAUNCH((dwconv3x3_kernel<S, A, 0, 5>), grid, block, 0, s, p); break; } \
        switch (g_dw_variant & 3) {                                                                 \
            /* stride 2 (the layer in front of a pointwise GEMM): the output leaves with the non-temporal hint -- in the network the   \
               GEMM behind it runs 5 us faster (57.5 -> 52.3 at 24 x 24 x 128 -> 256) and the layer itself 1-3 us; non-temporal LOADS \
               (variants 1, 3) cost the layer 13 us, the hint on the split-row form (variant bit 2) nothing either way */        \
            case 0: if (S == 2) HSEFR_LAUNCH((dwconv3x3_kernel<S, A, 2, 2>), grid, block, 0, s, p);                          \
                    else HSEFR_LAUNCH((dwconv3x3_kernel<S, A, 0, 2>), grid, block, 0, s, p);                                  \
                    break;                                                                                                            \
            case 1: HSEFR_LAUNCH((dwconv3x3_kernel<S, A, 1, 2>), grid, block, 0, s, p); break;   \
            case 2: HSEFR_LAUNCH((dwconv3x3_kernel<S, A, 2, 2>), grid, block, 0, s, p); break;   \
            default: HSEFR_LAUNCH((dwconv3x3_kernel<S, A, 3, 2>), grid, block, 0, s, p); break;  \
        }                                                                                           \
    } while (0)
    if (stride == 1) {
        if (act == HSEFR_ACT_RELU6) HSEFR_DW_LAUNCH(1, HSEFR_ACT_RELU6);
        else if (act == HSEFR_ACT_RELU) HSEFR_DW_LAUNCH(1, HSEFR_ACT_RELU);
        else if (act == HSEFR_ACT_NONE) HSEFR_DW_LAUNCH(1, HSEFR_ACT_NONE);
        else { set_error("dwconv3x3: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
    } else {
        if (act == HSEFR_ACT_RELU6) HSEFR_DW_LAUNCH(2, HSEFR_ACT_RELU6);
        else if (act == HSEFR_ACT_RELU) HSEFR_DW_LAUNCH(2, HSEFR_ACT_RELU);
        else if (act == HSEFR_ACT_NONE) HSEFR_DW_LAUNCH(2, HSEFR_ACT_NONE);
        else { set_error("dwconv3x3: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
    }
#undef HSEFR_DW_LAUNCH
    return launch_status("dwconv3x3");
}

int launch_dwconv3x3(const float* x, const float* wgt, const float* scale, const float* shift, float* y,
                     int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh, int ow, int act,
                     hipStream_t s) {
    return launch_dwconv3x3_impl(x, wgt, scale, shift, y, n, h, w, c, stride, pad_t, pad_l, oh, ow, act, 0, s);
}

int launch_dwconv3x3_split(const float* x, const float* wgt, const float* scale, const float* shift, void* y_split,
                           int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh, int ow, int act,
                           int a_log2, hipStream_t s) {
    HSEFR_REQUIRE(a_log2 > 0, HSEFR_ERR_INVALID, "dwconv3x3 (split output): a_log2=%d", a_log2);
    return launch_dwconv3x3_impl(x, wgt, scale, shift, (float*)y_split, n, h, w, c, stride, pad_t, pad_l, oh, ow, act, a_log2, s);
}

}  // namespace hsefr
