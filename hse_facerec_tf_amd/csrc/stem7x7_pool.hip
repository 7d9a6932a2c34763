// ResNet stem in ONE kernel: 7x7 / stride 2 / pad 3 convolution over the 3-channel fp32 image + scale + shift + ReLU
// -> 3x3 / stride 2 max-pool -> bf16 NHWC (conv1/7x7_s2 + BN + ReLU + pool1/3x3_s2 of resnet50_ft, the graph behind
// vgg2_resnet.pb at facerec_test.py:213).  gfx950.
//
// Why: layer by layer the 112 x 112 x 64 conv1 map is written (205 MB at batch 128) and read back by the pool, and the
// unfused stem gathers every 7x7x3 window from global memory tap by tap (12x read amplification through the TA): 238 us + 59 us
// of a 2.85 ms batch, for 30 GFLOP.  Here a workgroup owns a 7 x 7 tile of POOLED pixels:
//   * the 35 x 35 input window behind it is read once, coalesced, converted to bf16 and kept in LDS (8 KB);
//   * the 15 x 15 conv pixels the tile needs are an implicit GEMM straight off that window: for conv pixel (cy, cx) the 21
//     values (7 px x 3 ch) of kernel row dy are CONTIGUOUS in window row 2 cy + dy, so a B fragment (8 consecutive k) is one
//     4-byte-aligned 16-byte LDS read -- no im2col copy.  K = 7 rows x 24 (21 + 3 values that meet zero weights) = 168, padded
//     to 11 steps of v_mfma_f32_32x32x16_bf16 (the unfused kernel padded rows to 32: 16 steps);
//   * the weight fragments (64 channels x 176) stay in registers for the life of the persistent workgroup;
//   * conv results (scale, shift, ReLU, bf16) are parked in LDS, pooled from there (values are >= 0 after the ReLU, so bf16
//     bit patterns order like unsigned integers and pixels outside the map count as 0 -- the same result as a clipped window),
//     and leave as whole 128-byte rows.
// Rounding points are those of the two-kernel path: bf16 input, exact products, fp32 accumulation (in a different order: a
// few results differ by one bf16 ulp), bf16 after the ReLU; max-pooling commutes with the rounding.
#include "common.h"

namespace hsefr {

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

__device__ __forceinline__ u16 f2bf(float f) { return (u16)hsefr_bf16_bits(f); }      // round-to-nearest-even (common.h)

struct StemPoolParams {
    const float* x;      // [N,H,W,3] fp32
    const u16* wt;       // [64][8][32] bf16, k = dy*32 + dx*3 + ci, zero padded (resnet50.pack_stem_weight)
    const float* scale;  // [64]
    const float* shift;  // [64]
    u16* y;              // [N,PH,PW,64] bf16
    int H, W, OH, OW, PH, PW, ppt, ppl;
    unsigned tiles_x, tiles_per_img, tiles;
};

constexpr int TP = 7;              // pooled tile edge
constexpr int TC = 2 * TP + 1;     // conv tile edge (15)
constexpr int TWIN = 2 * TC + 5;   // input window edge (35)
constexpr int WROWS = TWIN + 1;    // + one zero row: the padding chunk (dy = 7) of the last conv row reads it
constexpr int WPITCH = 112;        // bf16 per window row: 105 used, the rest zero (read against zero weights)
constexpr int NLD = (TWIN * TWIN * 3 + 255) / 256;   // window floats per thread (15)
constexpr int NQ = TC * TC;        // 225 conv pixels, processed as 8 blocks of 32

struct __attribute__((packed, aligned(4))) Frag4 { bf16x8 v; };   // a 16-byte fragment at a 4-byte-aligned LDS address

__global__ __launch_bounds__(256, 2) void stem7x7_pool_bf16_kernel(StemPoolParams p) {
    __shared__ __attribute__((aligned(16))) u16 win[WROWS * WPITCH];        // 8 KB
    __shared__ __attribute__((aligned(16))) unsigned char Cs[256 * 128];    // conv tile [pixel][64 ch] bf16, chunk-swizzled (32 KB)
    __shared__ __attribute__((aligned(16))) float Es[128];                  // scale | shift
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    if (blockIdx.x >= p.tiles) return;

    for (int i = tid; i < WROWS * WPITCH / 2; i += 256) ((unsigned*)win)[i] = 0u;
    if (tid < 64) Es[tid] = p.scale[tid];
    else if (tid < 128) Es[tid] = p.shift[tid - 64];

    // weight fragments: channel block cb, step s -> 16 bytes at [channel][dy][8 j], chunk c = 2 s + lh = 3 dy + j
    bf16x8 wf[2][11];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int s = 0; s < 11; ++s) {
            const int c = 2 * s + lh, dy = c / 3, j = c - 3 * dy;
            wf[cb][s] = *(const bf16x8*)(p.wt + (size_t)(cb * 32 + li) * 256 + dy * 32 + 8 * j);
        }

    // window element owned by this thread in round i: (row, column) of the 35 x 105 float window
    int wrc[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        const int r = e / (TWIN * 3), c = e - r * (TWIN * 3);
        wrc[i] = e < TWIN * TWIN * 3 ? (r << 8 | c) : -1;
    }
    float g[NLD];
    int t_img = 0, t_py0 = 0, t_px0 = 0;
    auto decode = [&](unsigned t, int& img, int& py0, int& px0) {
        img = (int)(t / p.tiles_per_img);
        const unsigned rem = t - (unsigned)img * p.tiles_per_img;
        const unsigned ty = rem / p.tiles_x;
        py0 = (int)ty * TP;
        px0 = (int)(rem - ty * p.tiles_x) * TP;
    };
    auto gather = [&](unsigned t) {
        int img, py0, px0;
        decode(t, img, py0, px0);
        const int iy0 = 2 * (2 * py0 - p.ppt) - 3, gx0 = (2 * (2 * px0 - p.ppl) - 3) * 3;
        const float* im = p.x + (size_t)img * p.H * p.W * 3;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int r = wrc[i] >> 8, c = wrc[i] & 255;
            const int iy = iy0 + r, gx = gx0 + c;
            const bool ok = wrc[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)gx < (unsigned)(p.W * 3);
            g[i] = ok ? im[(size_t)iy * (p.W * 3) + gx] : 0.f;
        }
    };
    auto scatter = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (wrc[i] >= 0) win[(wrc[i] >> 8) * WPITCH + (wrc[i] & 255)] = f2bf(g[i]);
    };

    // per-lane fragment origin of the two pixel blocks this wave multiplies (blocks wave and wave + 4)
    int qbase[2], qcy[2], qcx[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        int q = 32 * (wave + 4 * b) + li;
        q = q < NQ ? q : NQ - 1;
        qcy[b] = q / TC;
        qcx[b] = q - qcy[b] * TC;
        qbase[b] = (2 * qcy[b] * WPITCH + 6 * qcx[b]) * 2;
    }

    unsigned t = blockIdx.x;
    gather(t);
    __syncthreads();           // zero fill and constants are in place
    while (true) {
        decode(t, t_img, t_py0, t_px0);
        scatter();
        __syncthreads();       // window complete; every wave is past the previous tile's pooling pass
        const unsigned tn = t + gridDim.x;
        const bool more = tn < p.tiles;
        if (more) gather(tn);
        const int cy0 = 2 * t_py0 - p.ppt, cx0 = 2 * t_px0 - p.ppl;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
            const unsigned char* wb = (const unsigned char*)win + qbase[b];
#pragma unroll
            for (int s = 0; s < 11; ++s) {
                const int c0 = 2 * s, c1 = 2 * s + 1;
                const int o0 = ((c0 / 3) * WPITCH + 8 * (c0 % 3)) * 2, o1 = ((c1 / 3) * WPITCH + 8 * (c1 % 3)) * 2;
                const bf16x8 xa = ((const Frag4*)(wb + (lh ? o1 : o0)))->v;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][s], xa, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][s], xa, acc1, 0, 0, 0);
            }
            // lane -> conv pixel q, registers 4 g .. 4 g + 3 -> channels cb * 32 + 8 g + 4 lh + (0..3); pixels outside the map -> 0
            const int q = 32 * (wave + 4 * b) + li;
            const bool inmap = (unsigned)(cy0 + qcy[b]) < (unsigned)p.OH && (unsigned)(cx0 + qcx[b]) < (unsigned)p.OW;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int ch = cb * 32 + 8 * gq + 4 * lh;
                    const float4 sc = *(const float4*)(Es + ch);
                    const float4 sh = *(const float4*)(Es + 64 + ch);
                    const f32x16& a = cb ? acc1 : acc0;
                    float f0 = fmaxf(fmaf(a[4 * gq], sc.x, sh.x), 0.f), f1 = fmaxf(fmaf(a[4 * gq + 1], sc.y, sh.y), 0.f);
                    float f2 = fmaxf(fmaf(a[4 * gq + 2], sc.z, sh.z), 0.f), f3 = fmaxf(fmaf(a[4 * gq + 3], sc.w, sh.w), 0.f);
                    ushort4 o;
                    o.x = inmap ? f2bf(f0) : (u16)0; o.y = inmap ? f2bf(f1) : (u16)0;
                    o.z = inmap ? f2bf(f2) : (u16)0; o.w = inmap ? f2bf(f3) : (u16)0;
                    *(ushort4*)(Cs + q * 128 + 16 * ((ch >> 3) ^ (q & 7)) + 2 * (ch & 7)) = o;
                }
        }
        __syncthreads();       // conv tile parked; the window is free for the next tile's scatter
        // pooling pass: item = (pooled pixel of the tile, 8-channel chunk); 49 x 8 items over 256 threads
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int item = tid + 256 * rr;
            if (item < TP * TP * 8) {
                const int pp = item >> 3, c8 = item & 7;
                const int pyl = pp / TP, pxl = pp - pyl * TP;
                const int py = t_py0 + pyl, px = t_px0 + pxl;
                if (py < p.PH && px < p.PW) {
                    u16x8 m = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const int q = (2 * pyl + dy) * TC + 2 * pxl + dx;
                            const u16x8 v = *(const u16x8*)(Cs + q * 128 + 16 * (c8 ^ (q & 7)));
                            m = __builtin_elementwise_max(m, v);
                        }
                    *(u16x8*)(p.y + (((size_t)t_img * p.PH + py) * p.PW + px) * 64 + c8 * 8) = m;
                }
            }
        }
        if (!more) break;
        t = tn;
    }
}

}  // namespace

int launch_stem7x7_pool_bf16(const float* x, const void* wt, const float* scale, const float* shift, void* y, int n, int h, int w,
                             int ph, int pw, int pool_pad_t, int pool_pad_l, hipStream_t s, const void* wfrag) {
    HSEFR_REQUIRE(n >= 0 && h > 0 && w > 0 && ph > 0 && pw > 0, HSEFR_ERR_INVALID, "stem7x7_pool: bad shape");
    HSEFR_REQUIRE((pool_pad_t == 0 || pool_pad_t == 1) && (pool_pad_l == 0 || pool_pad_l == 1), HSEFR_ERR_UNSUPPORTED,
                  "stem7x7_pool: pool padding (%d, %d) not in {0, 1}", pool_pad_t, pool_pad_l);
    const int oh = (h - 1) / 2 + 1, ow = (w - 1) / 2 + 1;      // 7x7 / 2, pad 3
    // every pooled pixel must see at least one conv pixel (its first window row / column lies inside the map)
    HSEFR_REQUIRE(2 * (ph - 1) - pool_pad_t < oh && 2 * (pw - 1) - pool_pad_l < ow, HSEFR_ERR_INVALID,
                  "stem7x7_pool: pooled size %dx%d does not fit a %dx%d conv map", ph, pw, oh, ow);
    if (n == 0) return HSEFR_OK;
    // the streaming form (stem7s_stream.hip, round 6) wherever it covers the shape; this patch kernel otherwise
    if (stem7s_stream_supported(n, h, w, ph, pw)) return launch_stem7s_stream(x, wt, scale, shift, y, n, h, w, ph, pw, pool_pad_t, pool_pad_l, s, wfrag);
    StemPoolParams p;
    p.x = x; p.wt = (const u16*)wt; p.scale = scale; p.shift = shift; p.y = (u16*)y;
    p.H = h; p.W = w; p.OH = oh; p.OW = ow; p.PH = ph; p.PW = pw; p.ppt = pool_pad_t; p.ppl = pool_pad_l;
    p.tiles_x = (unsigned)((pw + TP - 1) / TP);
    p.tiles_per_img = p.tiles_x * (unsigned)((ph + TP - 1) / TP);
    const long long tiles = (long long)n * p.tiles_per_img;
    HSEFR_REQUIRE(tiles < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "stem7x7_pool: too many tiles");
    p.tiles = (unsigned)tiles;
    const unsigned g = p.tiles < 512u ? p.tiles : 512u;
    HSEFR_LAUNCH(stem7x7_pool_bf16_kernel, dim3(g), dim3(256), 0, s, p);
    return launch_status("stem7x7_pool_bf16");
}

}  // namespace hsefr
