// bf16 convolution as an implicit GEMM on v_mfma_f32_32x32x16_bf16 (fp32 accumulate), NHWC, gfx950:
// the ResNet-50 (`resnet50_ft`, vgg2_resnet.pb at facerec_test.py:213) trunk -- 1x1 (stride 1|2) and
// 3x3 convolutions with folded BatchNorm (fp32 scale + shift in the epilogue), optional residual add
// and ReLU, bf16 activations in HBM.
//
//   Y[p, n] = act( scale[n] * sum_{kh,kw,c} X[pix(p) + (kh,kw), c] * Wt[n, (kh*KW+kw)*C + c] + shift[n] (+ R[p, n]) )
//
// Same skeleton as the fp32 pointwise GEMM (pwconv_f32.hip): 256 threads = 2x2 waves over a BM x BN
// tile, 128-B LDS rows (here 64 bf16 = one K-tile) with the (row>>1)&7 XOR swizzle, full-line 16-B
// staging copies, register-staged double buffering over the flattened (tap, channel-tile) K loop.
// Differences:
//  * the A tile is an im2col GATHER: a staging thread owns output pixels and, per K-tile, reads 16 B
//    (8 channels) of input pixel (oh*s+kh-pad, ow*s+kw-pad) from a clamped address; padding taps are
//    zeroed by a select (no branches);
//  * operands are swapped in the MFMA (A = weights, B = pixels) so a lane ends up with 4 CONSECUTIVE
//    output channels of one pixel per accumulator quad; the tile is written to LDS as packed bf16
//    (ds_write_b64) and leaves through a second, fully coalesced pass of 16-B row chunks that also
//    adds the residual and applies the activation -- no 2-byte scattered global stores.
#include <hip/hip_bf16.h>

#include "common.h"

namespace hsefr {

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

__device__ __forceinline__ u16 f2bf(float f) { return (u16)hsefr_bf16_bits(f); }      // round-to-nearest-even (common.h)
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float((unsigned)h << 16); }

constexpr int BKB = 64;  // bf16 elements per K-tile = one 128-B LDS row

__device__ __forceinline__ int swzb(int row, int chunk) { return row * 128 + 16 * (chunk ^ ((row >> 1) & 7)); }  // bytes

struct ConvParams {
    const u16* x;        // [N,H,W,C] bf16
    const u16* wt;       // [Cout][KH*KW*C] bf16
    const float* scale;  // [Cout]
    const float* shift;  // [Cout]
    const u16* res;      // [P,Cout] bf16 or null
    u16* y;              // [P,Cout] bf16
    int H, W, C, OH, OW, Cout, KH, KW, stride, pad_t, pad_l, act;
    unsigned P;          // N*OH*OW output pixels
    unsigned tiles_n, total_tiles;
    int reverse;         // sweep direction (common.h)
};

template <int BM, int BN, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_bf16_kernel(ConvParams p) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int AP = BM / 32, BP = BN / 32;
    constexpr int CROW = BN * 2 + 16;  // bytes per row of the C staging tile (padded: conflict-light b64 writes)
    constexpr int LDS_AB = 2 * (BM + BN) * 128;
    constexpr int LDS_C = BM * CROW;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_AB > LDS_C ? LDS_AB : LDS_C];
    unsigned char* As = smem;                 // [2][BM*128]
    unsigned char* Bs = smem + 2 * BM * 128;  // [2][BN*128]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int srow = tid >> 3, sch = tid & 7;

    const unsigned lt = xcd_remap_dir(blockIdx.x, p.total_tiles, p.reverse);
    const unsigned tn = lt % p.tiles_n, tm = lt / p.tiles_n;
    const unsigned m0 = tm * BM;
    const int n0 = tn * BN;
    const int Ktot = p.KH * p.KW * p.C;
    const int ctiles = p.C / BKB;
    const int KT = p.KH * p.KW * ctiles;

    // per staged row: clamped pixel decode (tail rows re-read the last pixel, never stored)
    int ih0[AP], iw0[AP];
    const u16* img[AP];
#pragma unroll
    for (int a = 0; a < AP; ++a) {
        unsigned pix = m0 + srow + 32 * a;
        if (pix > p.P - 1) pix = p.P - 1;
        const unsigned ow = pix % (unsigned)p.OW, t2 = pix / (unsigned)p.OW;
        const unsigned oh = t2 % (unsigned)p.OH, n = t2 / (unsigned)p.OH;
        ih0[a] = (int)oh * p.stride - p.pad_t;
        iw0[a] = (int)ow * p.stride - p.pad_l;
        img[a] = p.x + (size_t)n * p.H * p.W * p.C + 8 * sch;
    }
    const u16* bg = p.wt + (size_t)(n0 + srow) * Ktot + 8 * sch;

    bf16x8 ra[AP], rb[BP];
    auto gload = [&](int kt) {
        const int tap = kt / ctiles, c0 = (kt - tap * ctiles) * BKB;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int a = 0; a < AP; ++a) {
            const int ih = ih0[a] + kh, iw = iw0[a] + kw;
            const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
            const bf16x8 v = *(const bf16x8*)(img[a] + ((size_t)ihc * p.W + iwc) * p.C + c0);
            const bool ok = ih == ihc && iw == iwc;
            bf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = ok ? v[e] : (short)0;
            ra[a] = z;
        }
#pragma unroll
        for (int b = 0; b < BP; ++b) rb[b] = *(const bf16x8*)(bg + (size_t)32 * b * Ktot + (size_t)kt * BKB);
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int a = 0; a < AP; ++a) *(bf16x8*)(As + buf * BM * 128 + swzb(srow + 32 * a, sch)) = ra[a];
#pragma unroll
        for (int b = 0; b < BP; ++b) *(bf16x8*)(Bs + buf * BN * 128 + swzb(srow + 32 * b, sch)) = rb[b];
    };

    // acc[ni][mi]: rows = output channels (weights are the MFMA A operand), columns = pixels
    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    gload(0);
    // The residual tile does not depend on anything computed here: request it NOW (registers), so that its HBM round
    // trip overlaps the whole K loop instead of sitting between the last MFMA and the first store (the 1x1 "increase"
    // layers have one or two K-tiles and twice as many residual bytes as input bytes).
    constexpr int CH_PER_ROW = BN / 8;                 // 16-B chunks per tile row
    constexpr int NCH = BM * CH_PER_ROW / 256;         // chunks per thread
    bf16x8 rres[NCH];
    if (p.res) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int i = tid + 256 * j, r = i / CH_PER_ROW, c = i - r * CH_PER_ROW;
            unsigned pix = m0 + r;
            if (pix > p.P - 1) pix = p.P - 1;           // tail rows: a valid address, never stored
            rres[j] = *(const bf16x8*)(p.res + (size_t)pix * p.Cout + n0 + c * 8);
        }
    }
    swrite(0);
    __syncthreads();
    const int xrow = wm * WM + li, wrow = wn * WN + li;
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) gload(kt + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x8 xa[MI], wb[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xa[mi] = *(const bf16x8*)(As + cur * BM * 128 + swzb(xrow + mi * 32, 2 * q + lh));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) wb[ni] = *(const bf16x8*)(Bs + cur * BN * 128 + swzb(wrow + ni * 32, 2 * q + lh));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
        }
        if (kt + 1 < KT) swrite(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue 1: scale/shift in fp32, pack 4 consecutive channels, park the tile in LDS -------------
    // D[row = channel][col = pixel]: lane -> pixel li, registers 4g..4g+3 -> channels 8g + 4*lh + 0..3
    unsigned char* Cs = smem;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = wn * WN + ni * 32 + 8 * g + 4 * lh;  // within the tile
            const float4 sc = *(const float4*)(p.scale + n0 + ch);
            const float4 sh = *(const float4*)(p.shift + n0 + ch);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int prow = wm * WM + mi * 32 + li;
                ushort4 o;
                o.x = f2bf(fmaf(acc[ni][mi][4 * g + 0], sc.x, sh.x));
                o.y = f2bf(fmaf(acc[ni][mi][4 * g + 1], sc.y, sh.y));
                o.z = f2bf(fmaf(acc[ni][mi][4 * g + 2], sc.z, sh.z));
                o.w = f2bf(fmaf(acc[ni][mi][4 * g + 3], sc.w, sh.w));
                *(ushort4*)(Cs + prow * CROW + ch * 2) = o;
            }
        }
    }
    __syncthreads();
    // ---- epilogue 2: coalesced 16-B chunks, residual add + activation on the way out ---------------------
    const bool full_tile = m0 + BM <= p.P;             // full tiles store unconditionally (no per-store branch / vmcnt(0))
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int i = tid + 256 * j, r = i / CH_PER_ROW, c = i - r * CH_PER_ROW;
        const unsigned pix = m0 + r;
        bf16x8 v = *(const bf16x8*)(Cs + r * CROW + c * 16);
        if (p.res || p.act != HSEFR_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = bf2f((u16)v[e]);
                if (p.res) f = bf2f(f2bf(f)) + bf2f((u16)rres[j][e]);
                if (p.act == HSEFR_ACT_RELU) f = fmaxf(f, 0.f);
                else if (p.act == HSEFR_ACT_RELU6) f = fminf(fmaxf(f, 0.f), 6.f);
                v[e] = (short)f2bf(f);
            }
        }
        const size_t off = (size_t)pix * p.Cout + n0 + c * 8;
        if (full_tile) *(bf16x8*)(p.y + off) = v;
        else if (pix < p.P) *(bf16x8*)(p.y + off) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stem: 7x7 / stride 2 / pad 3 convolution over the 3-channel fp32 image -> bf16, + scale + shift + ReLU
// (conv1/7x7_s2 + BN + ReLU of resnet50_ft).  K = 7 rows x (7 px * 3 ch = 21 contiguous floats, padded to
// 32) = 224 -> 4 K-tiles of 64 (last half-tile zero).  A workgroup gathers 64 output pixels (4 threads
// per pixel, <= 2 kernel rows each) into bf16 LDS rows and multiplies by the [Cout=64][256] weight image.
struct StemParams {
    const float* x;      // [N,H,W,3] fp32
    const u16* wt;       // [64][256] bf16, k = dy*32 + dx*3 + ci (zero padded)
    const float* scale;
    const float* shift;
    u16* y;              // [P,64] bf16
    int H, W, OH, OW, act;
    unsigned P, tiles;
};

struct __attribute__((packed, aligned(4))) F3s { float a, b, c; };

__global__ __launch_bounds__(256, 2) void stem7x7_bf16_kernel(StemParams p) {
    constexpr int BM = 64, KP = 256;
    __shared__ __attribute__((aligned(16))) u16 As[BM * KP];   // [pixel][k]  (32 KB)
    __shared__ __attribute__((aligned(16))) u16 Bs[64 * KP];   // [cout][k]   (32 KB)
    __shared__ __attribute__((aligned(16))) u16 Cs[BM * 64];   // output tile [pixel][cout] for the coalesced store pass (8 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // rows are 512 B = 4 x 128-B K-tiles; chunk swizzle inside each 128-B tile
    // (rows are 512 B apart: every row starts on the same bank.  The chunk XOR spreads 8 row pairs over the 8 chunks of a
    // K-tile; the K-tile XOR puts the two rows of a pair in opposite halves of the 256-B bank window -- without it every
    // fragment read was a 2-way conflict: 61 % of the kernel's LDS cycles)
    auto off = [](int row, int kt, int chunk) { return row * 512 + (kt ^ (row & 1)) * 128 + 16 * (chunk ^ ((row >> 1) & 7)); };

    for (int i = tid; i < 64 * 32; i += 256) {   // weights: 64 rows x 32 chunks of 16 B
        const int r = i >> 5, c = i & 31;
        *(bf16x8*)((unsigned char*)Bs + off(r, c >> 3, c & 7)) = *(const bf16x8*)(p.wt + (size_t)r * KP + c * 8);
    }
    // gather: 4 threads per output pixel, kernel rows q and q+4 each (row 7 = the zero half-tile); one dwordx3 per tap
    // pixel from a clamped address, padding applied as 0/1 factors at scatter time.  The next tile's gather is requested
    // before the current tile's MFMAs and stores.
    const int pl = tid >> 2, q = tid & 3;
    F3s g[2][7];
    int gih0 = 0, giw0 = 0;
    auto gather = [&](unsigned t) {
        unsigned pix = t * BM + pl;
        if (pix > p.P - 1) pix = p.P - 1;
        const unsigned ow = pix % (unsigned)p.OW, t2 = pix / (unsigned)p.OW;
        const unsigned oh = t2 % (unsigned)p.OH, n = t2 / (unsigned)p.OH;
        gih0 = (int)oh * 2 - 3; giw0 = (int)ow * 2 - 3;
        const float* img = p.x + (size_t)n * p.H * p.W * 3;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int dy = q + 4 * rr;
            const int ihc = min(max(gih0 + (dy < 7 ? dy : 6), 0), p.H - 1);
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) {
                const int iwc = min(max(giw0 + dx, 0), p.W - 1);
                g[rr][dx] = *(const F3s*)(img + (unsigned)(ihc * p.W + iwc) * 3u);
            }
        }
    };
    auto scatter = [&]() {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int dy = q + 4 * rr;          // 0..7 (7 = the zero half-tile)
            const int ih = gih0 + dy;
            const bool rowok = dy < 7 && ih >= 0 && ih < p.H;
            float v[32];
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) {
                const int iw = giw0 + dx;
                const float m = (rowok && iw >= 0 && iw < p.W) ? 1.f : 0.f;
                v[dx * 3 + 0] = g[rr][dx].a * m;
                v[dx * 3 + 1] = g[rr][dx].b * m;
                v[dx * 3 + 2] = g[rr][dx].c * m;
            }
#pragma unroll
            for (int e = 21; e < 32; ++e) v[e] = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {       // 32 k-values = 4 chunks of the half K-tile (dy&1)
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(v[c * 8 + e]);
                *(bf16x8*)((unsigned char*)As + off(pl, dy >> 1, (dy & 1) * 4 + c)) = o;
            }
        }
    };
    unsigned t = blockIdx.x;
    if (t >= p.tiles) return;
    gather(t);
    scatter();
    __syncthreads();
    const int wm = wave >> 1, wn = wave & 1;
    while (true) {
        const unsigned tn = t + gridDim.x;
        const bool more = tn < p.tiles;
        if (more) gather(tn);
        // 64 pixels x 64 channels: wave -> (pixel half, channel half), one 32x32 block each
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const bf16x8 xa = *(const bf16x8*)((unsigned char*)As + off(wm * 32 + li, kt, 2 * qq + lh));
                const bf16x8 wb = *(const bf16x8*)((unsigned char*)Bs + off(wn * 32 + li, kt, 2 * qq + lh));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, xa, acc, 0, 0, 0);
            }
        // lane -> pixel wm*32 + li, registers 4g..4g+3 -> channels wn*32 + 8g + 4lh + (0..3): park as bf16, 8 B per write
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int ch = wn * 32 + 8 * gq + 4 * lh;
            const float4 sc = *(const float4*)(p.scale + ch);
            const float4 sh = *(const float4*)(p.shift + ch);
            float f[4] = {fmaf(acc[4 * gq], sc.x, sh.x), fmaf(acc[4 * gq + 1], sc.y, sh.y),
                          fmaf(acc[4 * gq + 2], sc.z, sh.z), fmaf(acc[4 * gq + 3], sc.w, sh.w)};
            if (p.act == HSEFR_ACT_RELU) { f[0] = fmaxf(f[0], 0.f); f[1] = fmaxf(f[1], 0.f); f[2] = fmaxf(f[2], 0.f); f[3] = fmaxf(f[3], 0.f); }
            ushort4 o;
            o.x = f2bf(f[0]); o.y = f2bf(f[1]); o.z = f2bf(f[2]); o.w = f2bf(f[3]);
            const int prow = wm * 32 + li;
            *(ushort4*)((unsigned char*)Cs + prow * 128 + 16 * ((ch >> 3) ^ (prow & 7)) + 2 * (ch & 7)) = o;
        }
        __syncthreads();     // tile parked; every wave is done reading As
        // coalesced pass: 64 rows x 8 chunks of 16 B = 2 chunks per thread, whole 128-B rows
        const bool full_tile = (unsigned long long)t * BM + BM <= (unsigned long long)p.P;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j, r = i >> 3, c = i & 7;
            const bf16x8 v = *(const bf16x8*)((unsigned char*)Cs + r * 128 + 16 * (c ^ (r & 7)));
            const unsigned pix = t * BM + r;
            if (full_tile) *(bf16x8*)(p.y + (size_t)pix * 64 + c * 8) = v;
            else if (pix < p.P) *(bf16x8*)(p.y + (size_t)pix * 64 + c * 8) = v;
        }
        if (!more) break;
        scatter();
        __syncthreads();     // next tile's rows are in As; Cs is free again
        t = tn;
    }
}

// 3x3 / stride 2 max-pool, bf16 NHWC, explicit top/left padding (0 = Caffe ceil-mode, window clipped at
// the bottom/right edge; TF SAME passes its own pads).  Thread = 8 channels of one output pixel.
__global__ __launch_bounds__(256) void maxpool3x3s2_bf16_kernel(const u16* __restrict__ x, u16* __restrict__ y, int H,
                                                                int W, int C8, int OH, int OW, int pad_t, int pad_l,
                                                                unsigned total) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const unsigned c8 = i % (unsigned)C8, t1 = i / (unsigned)C8;
    const unsigned ow = t1 % (unsigned)OW, t2 = t1 / (unsigned)OW;
    const unsigned oh = t2 % (unsigned)OH, n = t2 / (unsigned)OH;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int dy = 0; dy < 3; ++dy) {
        const int ih = (int)oh * 2 - pad_t + dy;
        if (ih < 0 || ih >= H) continue;
        for (int dx = 0; dx < 3; ++dx) {
            const int iw = (int)ow * 2 - pad_l + dx;
            if (iw < 0 || iw >= W) continue;
            const bf16x8 v = *(const bf16x8*)(x + (((size_t)n * H + ih) * W + iw) * C8 * 8 + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], bf2f((u16)v[e]));
        }
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(m[e]);
    *(bf16x8*)(y + (size_t)i * 8) = o;
}

// Global average pool of bf16 activations -> fp32 (pool5/7x7_s1): wave = 64 channels x 4 HW slices.
__global__ __launch_bounds__(256) void gap_bf16_kernel(const u16* __restrict__ x, float* __restrict__ y, int n, int hw, int c8) {
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int groups = (c8 + 7) / 8;  // 64-channel groups per image (8 lanes x 8 channels)
    if (wave_global >= n * groups) return;
    const int img = wave_global / groups, grp = wave_global - img * groups;
    const int cq = grp * 8 + (lane & 7), part = lane >> 3;  // 8 HW slices
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (cq < c8) {
        const u16* p = x + (size_t)img * hw * c8 * 8 + cq * 8;
        for (int i = part; i < hw; i += 8) {
            const bf16x8 v = *(const bf16x8*)(p + (size_t)i * c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f((u16)v[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += __shfl_xor(acc[e], 8);
        acc[e] += __shfl_xor(acc[e], 16);
        acc[e] += __shfl_xor(acc[e], 32);
    }
    if (part == 0 && cq < c8) {
        float* o = y + (size_t)img * c8 * 8 + cq * 8;
        const float d = (float)hw;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = acc[e] / d;
    }
}

struct TileCfgB { int bm, bn, occ; };

TileCfgB choose_tile_b(unsigned pixels, int cout) {
    const TileCfgB cands[3] = {{128, 128, 2}, {128, 64, 3}, {64, 64, 3}};
    const double eff[3] = {1.00, 0.92, 0.80};
    int best = -1;
    double best_cost = 0;
    for (int i = 0; i < 3; ++i) {
        if (cout % cands[i].bn) continue;
        const long long tiles = (long long)((pixels + cands[i].bm - 1) / cands[i].bm) * (cout / cands[i].bn);
        const long long slots = 256ll * cands[i].occ;
        const long long rounds = (tiles + slots - 1) / slots;
        const double cost = (double)rounds * cands[i].occ * cands[i].bm * cands[i].bn / eff[i];
        if (best < 0 || cost < best_cost) { best = i; best_cost = cost; }
    }
    return cands[best];
}

template <int BM, int BN, int OCC>
int launch_conv_cfg(ConvParams p, hipStream_t s) {
    p.tiles_n = p.Cout / BN;
    const long long total = (long long)((p.P + BM - 1) / BM) * p.tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv_bf16: too many tiles");
    p.total_tiles = (unsigned)total;
    p.reverse = sweep_reverse();
    HSEFR_LAUNCH((conv_bf16_kernel<BM, BN, OCC>), dim3((unsigned)total), dim3(256), 0, s, p);
    return launch_status("conv_bf16");
}

}  // namespace

int launch_conv_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y,
                     int n, int h, int w, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t,
                     int pad_l, int act, hipStream_t s) {
    HSEFR_REQUIRE(c > 0 && c % 64 == 0, HSEFR_ERR_UNSUPPORTED, "conv_bf16: cin=%d must be a multiple of 64", c);
    HSEFR_REQUIRE(cout > 0 && cout % 64 == 0, HSEFR_ERR_UNSUPPORTED, "conv_bf16: cout=%d must be a multiple of 64", cout);
    HSEFR_REQUIRE(kh >= 1 && kw >= 1 && kh <= 7 && kw <= 7 && stride >= 1, HSEFR_ERR_UNSUPPORTED, "conv_bf16: kernel %dx%d/%d", kh, kw, stride);
    HSEFR_REQUIRE(act == HSEFR_ACT_NONE || act == HSEFR_ACT_RELU || act == HSEFR_ACT_RELU6, HSEFR_ERR_UNSUPPORTED, "conv_bf16: act %d", act);
    HSEFR_REQUIRE(n >= 0 && h > 0 && w > 0 && oh > 0 && ow > 0, HSEFR_ERR_INVALID, "conv_bf16: bad shape");
    if (n == 0) return HSEFR_OK;
    const long long P = (long long)n * oh * ow;
    HSEFR_REQUIRE(P < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv_bf16: too many output pixels");
    // 1x1 (stride 1 | 2) on the four-wave LDS-DMA GEMM (conv1x1_w4_bf16.hip, round 5)
    if (kh == 1 && kw == 1 && pad_t == 0 && pad_l == 0 && (conv1x1_w4_forced() || conv1x1_w4_bf16_preferred(P, c, cout, res != nullptr)) &&
        conv1x1_w4_bf16_supported(n, h, w, c, oh, ow, cout, stride))
        return launch_conv1x1_w4_bf16(x, wt, scale, shift, res, y, n, h, w, c, oh, ow, cout, stride, act, s);
    // 3x3 / stride 1 / pad 1, window resident in LDS, four wide MFMA waves (conv3x3_w2_bf16.hip, round 5): the 56 / 28 / 14-pixel maps
    if (kh == 3 && kw == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && oh == h && ow == w && conv3x3_w2_bf16_supported(n, h, w, c, cout) &&
        (conv3x3_w2_bf16_preferred(h, w, cout) || conv3x3_w2_forced()))
        return launch_conv3x3_w2_bf16(x, wt, scale, shift, res, y, n, h, w, c, cout, act, s);
#ifdef HSEFR_DEV
    // development builds, only when forced ("w3_off" = 2): the first window 3x3 kernel (conv3x3_win_bf16.hip, round 2) -- conv3x3_w2_bf16.hip
    // took every 3x3 layer of ResNet-50 in round 5; maps wider than 64 pixels run on the kernels below in the product
    if (kh == 3 && kw == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && oh == h && ow == w && conv3x3_win_forced() &&
        conv3x3_win_bf16_supported(n, h, w, c, cout))
        return launch_conv3x3_win_bf16(x, wt, scale, shift, res, y, n, h, w, c, cout, act, s);
#endif
    // the LDS-DMA implicit GEMM (conv_dma_bf16.hip) where it measured faster at ResNet-50's shapes (tools/kbench_conv.py, batch 128):
    // the K-deep layers -- 3x3 convolutions up to ~150k output pixels (45 vs 63 us on the 14x14x256 layers), the stride-2
    // 1x1 layers from 256+ channels (a strided gather costs the DMA nothing: 57 vs 84 us on 56x56x256 -> 28x28x512), the 1x1 reductions from 1024+ channels.  The other 1x1 layers and the 64-channel stage stay
    // on the register-staged kernels (few K-steps per tile: the DMA pipeline's per-tile costs are not amortised, and the
    // residual is prefetched there).
    // (round 6: the stride-2 3x3 of the 56-pixel stage's last block -- 64 channels: one K-step per tap -- 34 us there, 27 on the general kernel)
    const bool deep = (kh * kw > 1 && P <= 150000 && !(stride == 2 && c <= 64)) || (kh * kw == 1 && stride == 2 && c >= 256) || (kh * kw == 1 && c >= 1024 && !res);
    if ((deep || conv_dma_forced()) && conv_dma_bf16_supported(n, h, w, c, oh, ow, cout, kh, kw))
        return launch_conv_dma_bf16(x, wt, scale, shift, res, y, n, h, w, c, oh, ow, cout, kh, kw, stride, pad_t, pad_l, act, s);
    ConvParams p;
    p.x = (const u16*)x; p.wt = (const u16*)wt; p.scale = scale; p.shift = shift; p.res = (const u16*)res; p.y = (u16*)y;
    p.H = h; p.W = w; p.C = c; p.OH = oh; p.OW = ow; p.Cout = cout; p.KH = kh; p.KW = kw; p.stride = stride;
    p.pad_t = pad_t; p.pad_l = pad_l; p.act = act; p.P = (unsigned)P;
    // 1x1 stride-1 layers (the bottlenecks' reduce / increase): the persistent, prefetching GEMM of conv1x1_bf16.hip
    if (kh == 1 && kw == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && oh == h && ow == w && conv1x1_bf16_enabled(res != nullptr, c, cout))
        return launch_conv1x1_bf16(x, wt, scale, shift, res, y, P, c, cout, act, s);
    const TileCfgB cfg = choose_tile_b(p.P, cout);
    if (cfg.bm == 128 && cfg.bn == 128) return launch_conv_cfg<128, 128, 2>(p, s);
    if (cfg.bm == 128 && cfg.bn == 64) return launch_conv_cfg<128, 64, 3>(p, s);
    return launch_conv_cfg<64, 64, 3>(p, s);
}

int launch_stem7x7_bf16(const float* x, const void* wt, const float* scale, const float* shift, void* y, int n, int h,
                        int w, int oh, int ow, int act, hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && h > 0 && w > 0 && oh > 0 && ow > 0, HSEFR_ERR_INVALID, "stem7x7: bad shape");
    if (n == 0) return HSEFR_OK;
    const long long P = (long long)n * oh * ow;
    HSEFR_REQUIRE(P < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "stem7x7: too many output pixels");
    StemParams p;
    p.x = x; p.wt = (const u16*)wt; p.scale = scale; p.shift = shift; p.y = (u16*)y;
    p.H = h; p.W = w; p.OH = oh; p.OW = ow; p.act = act; p.P = (unsigned)P; p.tiles = (unsigned)((P + 63) / 64);
    const unsigned g = p.tiles < 512u ? p.tiles : 512u;
    HSEFR_LAUNCH(stem7x7_bf16_kernel, dim3(g), dim3(256), 0, s, p);
    return launch_status("stem7x7_bf16");
}

int launch_maxpool3x3s2_bf16(const void* x, void* y, int n, int h, int w, int c, int oh, int ow, int pad_t, int pad_l,
                             hipStream_t s) {
    HSEFR_REQUIRE(c > 0 && c % 8 == 0, HSEFR_ERR_UNSUPPORTED, "maxpool: c=%d must be a multiple of 8", c);
    if (n == 0) return HSEFR_OK;
    const long long total = (long long)n * oh * ow * (c / 8);
    HSEFR_REQUIRE(total < (1ll << 32), HSEFR_ERR_UNSUPPORTED, "maxpool: too large");
    HSEFR_LAUNCH(maxpool3x3s2_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const u16*)x,
                       (u16*)y, h, w, c / 8, oh, ow, pad_t, pad_l, (unsigned)total);
    return launch_status("maxpool3x3s2_bf16");
}

int launch_gap_bf16(const void* x, float* y, int n, int hw, int c, hipStream_t s) {
    HSEFR_REQUIRE(c > 0 && c % 8 == 0, HSEFR_ERR_UNSUPPORTED, "gap_bf16: c=%d must be a multiple of 8", c);
    if (n == 0) return HSEFR_OK;
    const int c8 = c / 8;
    const long long waves = (long long)n * ((c8 + 7) / 8);
    HSEFR_LAUNCH(gap_bf16_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, (const u16*)x, y, n, hw, c8);
    return launch_status("gap_bf16");
}

}  // namespace hsefr
