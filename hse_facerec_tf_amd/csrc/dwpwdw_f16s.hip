// Fused [depthwise 3x3 stride 1 -> pointwise 1x1 (split-f16 products) -> depthwise 3x3 stride 2], NHWC fp32, gfx950.
//
// Replaces one whole stride-1 MobileNet block AND the depthwise half of the stride-2 block behind it, e.g. graph nodes
// conv_dw_3 ... conv_pw_3_relu ... conv_dw_4_relu of the frozen graph run by tf_sess.run at facerec_test.py:120 /
// facial_analysis.py:109.  Unfused, the block's 48x48x128 output (302 MB at batch 256) is written once and read once only
// to be decimated 4:1 by the next depthwise; here neither it nor the first depthwise's result ever leaves the CU:
// HBM traffic 302 MB in + 75 MB out instead of 604 + 377.
//
// Structure = dwpw3_f16s_kernel (csrc/dwpw_f16s.hip: LDS-DMA halo ring with zero fill from the buffer bounds check,
// weight ring, producer / consumer wave specialisation, one barrier per 32-channel chunk) with the patch turned around:
// a workgroup owns 3 x 6 pixels of the FINAL (stride-2) map.  They need the block's output on a 7 x 13 region (91 pixels:
// 96 GEMM rows), which needs the first depthwise on the same region, i.e. a 9 x 15 input halo.  Per patch:
//   steps 0..KT-1   producers: depthwise 1 of chunk k from the halo -> A tile (f16 hi/lo);  consumers: MFMAs of chunk k-1
//   then            consumers: acc * descale + shift, activation, ZERO for region pixels outside the map (= the second
//                   depthwise's padding) -> the 7 x 13 x 128 region `P` in LDS (fp32)
//   one step later  consumers: depthwise 2 (stride 2) from P -> scale, shift, ReLU6 -> 512-B-per-pixel stores
// while both roles are already working on the next patch.  The region is recomputed with a 1.26x overlap between
// neighbouring patches (91 region pixels per 18 outputs instead of 72) -- the price of never storing it.
// Same operation order as dwconv.hip / pwconv_f16s.hip throughout: bit-identical to the three kernels it replaces.
#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct B3Params {
    const float* x;          // [N,H,W,C]
    const f32x4* wd;         // depthwise 1 [9][C/4]
    const f32x4* dscale;     // [C/4]
    const f32x4* dshift;     // [C/4]
    const float* wsplit;     // pointwise split rows [128][C/32][64 f16]
    const float* descale;    // [128]
    const float* pshift;     // [128]
    const f32x4* wd2;        // depthwise 2 [9][32]
    const f32x4* d2scale;    // [32]
    const f32x4* d2shift;    // [32]
    float* y;                // [N,OH,OW,128]
    int H, W, C4, KT, OH, OW, pad_t2, pad_l2, tiles_w, tiles_h, nimg, reverse;
    unsigned long long* stamps;   // diagnostic builds (-DHSEFR_STEM_STAMPS) only
    unsigned total;
    float a_scale;
};

constexpr int ROWB = 128;
constexpr int FH = 3, FW = 6;                                   // final patch
constexpr int R1H = 2 * FH + 1, R1W = 2 * FW + 1, R1PIX = R1H * R1W;     // 7 x 13 = 91 region pixels, 96 GEMM rows
constexpr int R0W = R1W + 2, R0PIX = (R1H + 2) * R0W;           // 9 x 15 = 135 halo pixels
constexpr int HPIECES = (R0PIX + 7) / 8;                        // 17 DMA pieces of 8 pixels
constexpr int HSLOT = HPIECES * 1024;
constexpr int MROWS = 96;
constexpr int PP = 132;                                         // floats per region pixel in LDS (128 + 4: rows 4 banks apart)
constexpr int COUT = 128;
constexpr int B_ST = COUT * ROWB;
constexpr int B_OFF = 2 * HSLOT, A_OFF = B_OFF + 2 * B_ST, P_OFF = A_OFF + 2 * MROWS * ROWB;
constexpr int W_OFF = P_OFF + ((R1PIX * PP * 4 + 1023) / 1024) * 1024, E_OFF = W_OFF + 11 * 128 * 4, SMEM = E_OFF + 2 * COUT * 4;
static_assert(SMEM <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int swzb(int row, int chunk) { return row * ROWB + 16 * (chunk ^ ((row >> 1) & 7) ^ ((row & 1) << 2)); }
__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

template <int ACT>
__global__ __launch_bounds__(512, 2) void dwpwdw_f16s_kernel(B3Params p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C4 * 4, KT = p.KT;
    {   // resident constants: depthwise-1 taps [9][C], scale, shift; pointwise descale, shift
        f32x4* wl = (f32x4*)(smem + W_OFF);
        for (int i = tid; i < 9 * p.C4; i += 512) wl[i] = p.wd[i];
        for (int i = tid; i < p.C4; i += 512) { wl[9 * p.C4 + i] = p.dscale[i]; wl[10 * p.C4 + i] = p.dshift[i]; }
        float* el = (float*)(smem + E_OFF);
        for (int i = tid; i < COUT; i += 512) { el[i] = p.descale[i]; el[COUT + i] = p.pshift[i]; }
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (long long)p.nimg * p.H * p.W * C * 4);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.wsplit, (long long)COUT * C * 4);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y, (long long)p.nimg * p.OH * p.OW * COUT * 4);
    const unsigned OOB = 0xFFFFFFF0u;
    const unsigned nitem = (p.total - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int nsteps = (int)(nitem * KT);
    struct Item { int n, oh0, ow0; };
    auto decode = [&](unsigned i) {
        const unsigned lt = xcd_remap_dir(blockIdx.x + (i < nitem ? i : nitem - 1) * gridDim.x, p.total, p.reverse);
        Item it;
        it.ow0 = (lt % p.tiles_w) * FW;
        it.oh0 = ((lt / p.tiles_w) % p.tiles_h) * FH;
        it.n = lt / (p.tiles_w * p.tiles_h);
        return it;
    };
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    auto piece = [&](const __amdgpu_buffer_rsrc_t& r, unsigned lds_addr, unsigned voff, unsigned soff) {
        // issued from asm: hipcc would otherwise drain vmcnt before every later LDS read (csrc/dwpw_f16s.hip)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff),
                     "s"(r), "s"(__builtin_amdgcn_readfirstlane(soff))
                     : "memory");
    };

    STEM_STAMP_DECL;
    if (wave < 4) {
        // ======================= producers: halo DMA, depthwise 1 -> A tile =======================
        constexpr int HPW = 5;                          // pieces 5w .. 5w+4 (17 in all)
        unsigned hv[HPW];
        unsigned pf_i = 0;
        int pf_kc = 0, pf_step = 0;
        auto setup_halo = [&](unsigned i) {
            const Item it = decode(i);
            const int y0 = 2 * it.oh0 - p.pad_t2 - 1, x0 = 2 * it.ow0 - p.pad_l2 - 1;     // halo origin in the block's map
#pragma unroll
            for (int j = 0; j < HPW; ++j) {
                const int q = (wave * HPW + j) * 8 + (lane >> 3);
                const int hr = q / R0W, hc = q - hr * R0W;
                const int ih = y0 + hr, iw = x0 + hc;
                const bool ok = q < R0PIX && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                hv[j] = ok ? ((unsigned)((it.n * p.H + ih) * p.W + iw) * (unsigned)C + 4u * (lane & 7)) * 4u : OOB;
            }
        };
        auto halo_dma = [&]() {
            const unsigned base = lds0 + (pf_step & 1) * HSLOT;
            const unsigned so = (unsigned)pf_kc * 128u;
#pragma unroll
            for (int j = 0; j < HPW; ++j)
                if (wave * HPW + j < HPIECES) piece(rx, base + (wave * HPW + j) * 1024, hv[j], so);
            ++pf_step;
            if (++pf_kc == KT) {
                pf_kc = 0;
                setup_halo(++pf_i);
            }
        };
        // depthwise-1 work of this thread: channel quad tid & 7, region column (tid >> 3) & 15 (13 live), rows 0..3 or 4..6:
        // a 6 x 3 (5 x 3) window of the halo in registers, as in dwconv.hip -- 18 reads for 4 outputs
        const int quad = tid & 7, col = (tid >> 3) & 15, rg = tid >> 7;
        const bool col_live = col < R1W;
        const int r0 = 4 * rg, nrow = rg ? 3 : 4;
        const int hoff0 = ((r0 * R0W) + (col_live ? col : 0)) * 128 + quad * 16;
        setup_halo(0);
        if (nsteps > 0) halo_dma();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int dkc = 0;
        for (int g = -1; g <= nsteps + 1; ++g) {
            if (pf_step < nsteps) halo_dma();                         // halo of step g + 2
            STEM_STAMP(1);
            if (g >= KT + 1 && (g - 1) % KT == 0) STEM_STAMP_COUNT;
            if (g + 1 < nsteps) {
                const f32x4* wl = (const f32x4*)(smem + W_OFF) + dkc * 8 + quad;
                f32x4 wk[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) wk[i] = wl[i * p.C4];
                const f32x4 dsc = wl[9 * p.C4], dsh = wl[10 * p.C4];
                const unsigned char* hs = smem + ((g + 1) & 1) * HSLOT;
                unsigned char* At = smem + A_OFF + ((g + 1) & 1) * (MROWS * ROWB);
                const unsigned char* hp = hs + hoff0;
                f32x4 hwin[6][3];
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        if (a < 5 || rg == 0) hwin[a][b] = *(const f32x4*)(hp + (a * R0W + b) * 128);      // (rows 4..6 need 5 halo rows)
                auto row_sum = [&](int a, int b) {
                    f32x4 t = hwin[a][0] * wk[b];
                    t = __builtin_elementwise_fma(hwin[a][1], wk[b + 1], t);
                    return __builtin_elementwise_fma(hwin[a][2], wk[b + 2], t);
                };
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j >= nrow) continue;
                    const f32x4 o = __builtin_elementwise_fma((row_sum(j, 0) + row_sum(j + 1, 3)) + row_sum(j + 2, 6), dsc, dsh);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = relu6(o[e]);
                    v = v * p.a_scale;
                    const f16x4 hi = __builtin_convertvector(v, f16x4);
                    const f16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), f16x4);
                    const int R = (r0 + j) * R1W + col;
                    if (col_live) {
                        *(f16x4*)(At + swzb(R, quad >> 1) + 8 * (quad & 1)) = hi;
                        *(f16x4*)(At + swzb(R, 4 + (quad >> 1)) + 8 * (quad & 1)) = lo;
                    }
                }
                if (++dkc == KT) dkc = 0;
            }
            STEM_STAMP(2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the halo of step g + 2 has landed
            STEM_STAMP(0);
            __syncthreads();
            STEM_STAMP(3);
        }
        STEM_STAMP_FLUSH(p.stamps, lane, wave);
    } else {
        // ======================= consumers: weight DMA, MFMA, block output -> region P, depthwise 2 -> global =======================
        const int cw = wave - 4;                         // owns output channels 32 cw .. 32 cw + 31, all 96 rows
        const int li = lane & 31, lh = lane >> 5;
        const int brow = cw * 32 + li;
        unsigned bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (cw * 4 + j) * 8 + (lane >> 3);
            bv[j] = ((unsigned)r * (unsigned)C + 4u * ((lane & 7) ^ ((r >> 1) & 7) ^ ((r & 1) << 2))) * 4u;
        }
        int pb_kc = 0, pb_step = 0;
        auto b_dma = [&]() {
            const unsigned base = lds0 + B_OFF + (pb_step & 1) * B_ST;
#pragma unroll
            for (int j = 0; j < 4; ++j) piece(rw, base + (cw * 4 + j) * 1024, bv[j], (unsigned)pb_kc * 128u);
            ++pb_step;
            if (++pb_kc == KT) pb_kc = 0;
        };
        f32x16 acc[3];
        auto zero_acc = [&]() {
#pragma unroll
            for (int mi = 0; mi < 3; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
        };
        zero_acc();
        unsigned ci = 0;
        int ckc = 0;
        bool pending = false;
        __syncthreads();                                  // (the producers' prologue barrier; it also publishes the constants)
        // lane (li, lh) holds row m = li, columns 4 lh + 8 j + e of its 32 x 32 blocks (operands swapped): 4 consecutive channels
        f32x4 ds[4], sh[4];
        {
            const float* el = (const float*)(smem + E_OFF);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ds[j] = *(const f32x4*)(el + cw * 32 + 8 * j + 4 * lh);
                sh[j] = *(const f32x4*)(el + COUT + cw * 32 + 8 * j + 4 * lh);
            }
        }
        // depthwise-2 work items: output pixel (tc >> 5) + 8 j (18 live), channel quad tid & 31; its constants live in registers
        const int tc = tid - 256, q32 = tc & 31;
        f32x4 w2[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) w2[i] = p.wd2[i * 32 + q32];
        const f32x4 sc2 = p.d2scale[q32], sh2 = p.d2shift[q32];
        int poff[3], opx[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            opx[j] = (tc >> 5) + 8 * j;
            const int oc = opx[j] < FH * FW ? opx[j] : 0;
            poff[j] = P_OFF + (((2 * (oc / FW)) * R1W + 2 * (oc % FW)) * PP + q32 * 4) * 4;
        }
        auto depthwise2 = [&](unsigned item) {
            const Item it = decode(item);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int oh = it.oh0 + opx[j] / FW, ow = it.ow0 + opx[j] % FW;
                if (opx[j] >= FH * FW || oh >= p.OH || ow >= p.OW) continue;
                const unsigned char* ps = smem + poff[j];
                f32x4 rs[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {        // row sums left to right, rows top to bottom: the order of dwconv.hip
                    const unsigned char* pr = ps + dy * (R1W * PP * 4);
                    f32x4 t = *(const f32x4*)pr * w2[3 * dy];
                    t = __builtin_elementwise_fma(*(const f32x4*)(pr + PP * 4), w2[3 * dy + 1], t);
                    rs[dy] = __builtin_elementwise_fma(*(const f32x4*)(pr + 2 * PP * 4), w2[3 * dy + 2], t);
                }
                const f32x4 o = __builtin_elementwise_fma((rs[0] + rs[1]) + rs[2], sc2, sh2);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = relu6(o[e]);
                bstore16_welded(v, ry, ((unsigned)((it.n * p.OH + oh) * p.OW + ow) * (unsigned)COUT + 4u * q32) * 4u, 0u);
            }
        };

        for (int g = -1; g <= nsteps + 1; ++g) {
            if (g >= KT + 1 && (g - 1) % KT == 0) depthwise2((unsigned)((g - 1) / KT - 1));     // from the region written one step ago
            if (pending) {
                const Item it = decode(ci++);
                const int y10 = 2 * it.oh0 - p.pad_t2, x10 = 2 * it.ow0 - p.pad_l2;
#pragma unroll
                for (int mi = 0; mi < 3; ++mi) {
                    const int px = mi * 32 + li;
                    if (px < R1PIX) {
                        const int r = px / R1W, c = px - r * R1W;
                        const bool in = y10 + r >= 0 && y10 + r < p.H && x10 + c >= 0 && x10 + c < p.W;
                        unsigned char* pp = smem + P_OFF + (px * PP + cw * 32 + 4 * lh) * 4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = in ? apply_act<ACT>(fmaf(acc[mi][4 * j + e], ds[j][e], sh[j][e])) : 0.f;
                            *(f32x4*)(pp + 32 * j) = v;
                        }
                    }
                }
                zero_acc();
                pending = false;
                STEM_STAMP_COUNT;
            }
            STEM_STAMP(5);
            if (pb_step < nsteps) b_dma();                // weights of step g + 1
            STEM_STAMP(1);
            if (g >= 0 && g < nsteps) {
                const unsigned char* As = smem + A_OFF + (g & 1) * (MROWS * ROWB);
                const unsigned char* Bs = smem + B_OFF + (g & 1) * B_ST;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f16x8 ah[3], al[3];
#pragma unroll
                    for (int mi = 0; mi < 3; ++mi) {
                        ah[mi] = *(const f16x8*)(As + swzb(mi * 32 + li, 2 * s + lh));
                        al[mi] = *(const f16x8*)(As + swzb(mi * 32 + li, 4 + 2 * s + lh));
                    }
                    const f16x8 bh = *(const f16x8*)(Bs + swzb(brow, 2 * s + lh));
                    const f16x8 bl = *(const f16x8*)(Bs + swzb(brow, 4 + 2 * s + lh));
#pragma unroll
                    for (int mi = 0; mi < 3; ++mi) {
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al[mi], acc[mi], 0, 0, 0);
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah[mi], acc[mi], 0, 0, 0);
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah[mi], acc[mi], 0, 0, 0);
                    }
                }
                if (++ckc == KT) { ckc = 0; pending = true; }
            }
            STEM_STAMP(4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STEM_STAMP(0);
            __syncthreads();
            STEM_STAMP(3);
        }
#ifdef HSEFR_STEM_STAMPS
        if (lane == 0 && p.stamps && blockIdx.x < 256) {      // consumer waves report in the upper half of the stamp table
            unsigned long long* o = p.stamps + ((blockIdx.x + 256) * 4 + cw) * 10;
            for (int i_ = 0; i_ < 8; ++i_) o[i_] = st[i_];
            o[8] = __builtin_amdgcn_s_memtime() - tstart;
            o[9] = npatch;
        }
#endif
    }
}

}  // namespace

bool dwpwdw_f16s_supported(int c, int cout, int act2) {
    return c >= 64 && c <= 128 && c % 32 == 0 && cout == COUT && act2 == HSEFR_ACT_RELU6;
}

int launch_dwpwdw_f16s(const float* x, const float* wd, const float* dscale, const float* dshift, const void* wsplit,
                       const float* descale, const float* pshift, const float* wd2, const float* d2scale, const float* d2shift,
                       float* y, int n, int h, int w, int c, int cout, int pad_t2, int pad_l2, int oh2, int ow2, int a_log2,
                       int act, int act2, hipStream_t s) {
    HSEFR_REQUIRE(dwpwdw_f16s_supported(c, cout, act2), HSEFR_ERR_UNSUPPORTED,
                  "dwpwdw_f16split: c=%d cout=%d act2=%d not covered (c in 64..128 step 32, cout 128, ReLU6 after the second depthwise)", c,
                  cout, act2);
    HSEFR_REQUIRE(n >= 0 && h > 0 && w > 0 && oh2 > 0 && ow2 > 0 && (pad_t2 == 0 || pad_t2 == 1) && (pad_l2 == 0 || pad_l2 == 1),
                  HSEFR_ERR_INVALID, "dwpwdw_f16split: bad shape");
    HSEFR_REQUIRE(a_log2 > 0 && a_log2 <= 12, HSEFR_ERR_INVALID, "dwpwdw_f16split: a_log2=%d (the depthwise result is in [0,6]: 1..12)", a_log2);
    HSEFR_REQUIRE((long long)n * h * w * c * 4 < (1ll << 32) - 16 && (long long)n * oh2 * ow2 * cout * 4 < (1ll << 32) - 16,
                  HSEFR_ERR_UNSUPPORTED, "dwpwdw_f16split: tensors of 4 GiB or more");
    if (n == 0) return HSEFR_OK;
    B3Params p;
    p.x = x; p.wd = (const f32x4*)wd; p.dscale = (const f32x4*)dscale; p.dshift = (const f32x4*)dshift;
    p.wsplit = (const float*)wsplit; p.descale = descale; p.pshift = pshift;
    p.wd2 = (const f32x4*)wd2; p.d2scale = (const f32x4*)d2scale; p.d2shift = (const f32x4*)d2shift; p.y = y;
    p.H = h; p.W = w; p.C4 = c / 4; p.KT = c / 32; p.OH = oh2; p.OW = ow2; p.pad_t2 = pad_t2; p.pad_l2 = pad_l2;
    p.tiles_w = (ow2 + FW - 1) / FW; p.tiles_h = (oh2 + FH - 1) / FH;
    p.nimg = n; p.reverse = sweep_reverse();
    p.a_scale = ldexpf(1.f, a_log2);
    p.stamps = nullptr;
#ifdef HSEFR_STEM_STAMPS
    p.stamps = stamp_buffer(s);
#endif
    const long long total = (long long)n * p.tiles_w * p.tiles_h;
    HSEFR_REQUIRE(total < (1ll << 27), HSEFR_ERR_UNSUPPORTED, "dwpwdw_f16split: grid too large");
    p.total = (unsigned)total;
    const unsigned g = p.total < 256u ? p.total : 256u;     // one 8-wave workgroup (147 KB of LDS) per CU
#define HSEFR_B3(A) hipLaunchKernelGGL((dwpwdw_f16s_kernel<A>), dim3(g), dim3(512), 0, s, p)
    if (act == HSEFR_ACT_RELU6) HSEFR_B3(HSEFR_ACT_RELU6);
    else if (act == HSEFR_ACT_RELU) HSEFR_B3(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_NONE) HSEFR_B3(HSEFR_ACT_NONE);
    else { set_error("dwpwdw_f16split: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_B3
    return launch_status("dwpwdw_f16split");
}

}  // namespace hsefr
