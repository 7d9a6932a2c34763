// Fused depthwise 3x3 (+scale+shift+ReLU6) -> pointwise 1x1 (+shift+act) for ANY channel count, the pointwise products
// on the f16 MFMA from two-term f16 splits (fp32-grade; arithmetic and weight image: pwconv_f16s.hip), NHWC fp32, gfx950.
//
// Replaces one whole MobileNet block of the frozen graph (nodes DepthwiseConv2dNative, Mul, Add, Relu, Minimum, Maximum,
// Conv2D 1x1, Add, Relu, Minimum, Maximum -- e.g. #35-#49 -- run by tf_sess.run at facerec_test.py:120 /
// facial_analysis.py:109).  Unfused, the depthwise output makes a round trip through HBM (written by one kernel, read
// back by the GEMM): on the 96/48/24-pixel maps that is as many bytes as the block's real input and output together.
// Here it never leaves the CU.  The depthwise result is always in [0, 6] (ReLU6), which is exactly the bound the f16
// split needs, so this kernel has no fp32-MFMA twin.
//
// One workgroup (256 threads, 4 waves as 2 x 2) owns a patch of 128 output pixels (TH x TW = 8 x 16 or 16 x 8) and BN
// output channels, and walks the input channels in chunks of 32 (= one K-tile of the GEMM):
//   produce   every thread computes the depthwise result of 4 output rows x 1 column x 4 channels of the chunk with the
//             sliding-window scheme of dwconv.hip (coalesced float4 loads from clamped addresses, padding folded into
//             zeroed tap weights / 0-1 row factors, rows requested two iterations ahead), scales it by 2^12, splits it
//             into f16 hi + lo and keeps the 4 results in registers; the chunk's slice of the split weight rows
//             (BN x 128 B) is fetched alongside;
//   publish   barrier (the previous chunk's MFMA reads are done) -> A and B tiles written to LDS in the swizzled
//             128-B-row layout of pwconv_f16s.hip -> barrier;
//   contract  24 (BN = 128) v_mfma_f32_32x32x16_f16 per wave: acc += al*bh + ah*bl + ah*bh.
// After the last chunk: acc * descale + shift, activation, full 128-B row stores.  Two to three co-resident workgroups
// per CU interleave their produce / contract phases, which overlaps the streaming with the MFMAs.
// HBM traffic per patch: (TH*s+2) x (TW*s+2) x C in (halo re-reads hit L2; workgroup ids are XCD-remapped so that
// neighbouring patches and the N-tiles of one patch share an L2) + 128 x Cout out.
#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct DwPwSParams {
    const float4* x;       // [N,H,W,C]
    const float4* wd;      // depthwise [9][C/4]
    const float4* dscale;  // [C/4]
    const float4* dshift;  // [C/4]
    const float* wsplit;   // pointwise split rows [Cout][C/32][64 f16]  (byte-compatible with fp32 [Cout][C])
    const float* descale;  // [Cout]
    const float* pshift;   // [Cout]
    float* y;              // [N,OH,OW,Cout]
    int H, W, C4, KT, OH, OW, Cout, pad_t, pad_l, tiles_w, tiles_h, tiles_n;
    unsigned total;        // N * tiles_h * tiles_w * tiles_n work items
    float a_scale;         // 2^a_log2
    int reverse;           // sweep direction (common.h)
    int nimg;              // batch (LDS-DMA form: buffer resource sizes)
    unsigned m_n, m_w, m_h;   // v3: ceil(2^32 / d) for d = tiles_n, tiles_w, tiles_h (exact quotients for x * d < 2^32; 0 for d = 1)
    unsigned long long* stamps;   // diagnostic builds (-DHSEFR_STEM_STAMPS) only
};

constexpr int ROWB = 128;
__device__ __forceinline__ int swzb(int row, int chunk) { return row * ROWB + 16 * (chunk ^ ((row >> 1) & 7) ^ ((row & 1) << 2)); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float relu6(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
// x / d by multiplication with m = ceil(2^32 / d) (m = 0 stands for d = 1): exact while x * d < 2^32
__device__ __forceinline__ unsigned fastdiv(unsigned x, unsigned m) { return m ? __umulhi(x, m) : x; }

template <int STRIDE, int TW, int BN, int OCC, int ACT>
__global__ __launch_bounds__(256, OCC) void dwpw_f16s_kernel(DwPwSParams p) {
    constexpr int TH = 128 / TW;
    constexpr int U = TW * 8;             // (column, channel-quad) work items per patch row and chunk: 128 or 64
    constexpr int RG = 256 / U;           // row groups: 2 or 4
    constexpr int RPT = TH / RG;          // output rows per thread: 4
    static_assert(RPT == 4, "patch shapes 8x16 and 16x8");
    constexpr int WN = BN / 2, NI = WN / 32, BP = BN / 32;
    __shared__ __attribute__((aligned(16))) unsigned char As[128 * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[BN * ROWB];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int srow = tid >> 3, skq = tid & 7;
    // producer coordinates inside the patch
    const int u = tid % U, tw = u >> 3, c4l = u & 7;
    const int row0 = (tid / U) * RPT;
    const int arow = wm * 64 + li, brow = wn * WN + li;

    for (unsigned t = blockIdx.x; t < p.total; t += gridDim.x) {
        const unsigned lt = xcd_remap_dir(t, p.total, p.reverse);
        const int n0 = (lt % p.tiles_n) * BN;
        const unsigned pt = lt / p.tiles_n;
        const int ow0 = (pt % p.tiles_w) * TW;
        const int oh0 = ((pt / p.tiles_w) % p.tiles_h) * TH;
        const int n = pt / (p.tiles_w * p.tiles_h);

        // per-patch producer constants: clamped tap columns and their 0/1 padding masks
        const int ow = min(ow0 + tw, p.OW - 1);            // clamped: out-of-range columns are computed, never stored
        const int iw0 = ow * STRIDE - p.pad_l;
        const float ml = iw0 >= 0 ? 1.f : 0.f, mm = (iw0 + 1 >= 0 && iw0 + 1 < p.W) ? 1.f : 0.f, mr = iw0 + 2 < p.W ? 1.f : 0.f;
        const int cl = max(iw0, 0) * p.C4, cm = min(max(iw0 + 1, 0), p.W - 1) * p.C4, cr = min(iw0 + 2, p.W - 1) * p.C4;
        const float4* ximg = p.x + (size_t)n * p.H * p.W * p.C4;
        const int ih0 = (oh0 + row0) * STRIDE - p.pad_t;
        const float* bsrc = p.wsplit + (size_t)(n0 + srow) * p.KT * 32 + 4 * skq;

        f32x16 acc[2][NI];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        for (int kc = 0; kc < p.KT; ++kc) {
            // ---- produce: B slice loads first (longest latency to hide), then the depthwise chunk ----
            f32x4 rb[BP];
#pragma unroll
            for (int q = 0; q < BP; ++q) rb[q] = *(const f32x4*)(bsrc + (size_t)32 * q * p.KT * 32 + kc * 32);
            const int c4 = kc * 8 + c4l;
            const float4* xin = ximg + c4;
            float4 wk[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const float m = (i % 3 == 0) ? ml : (i % 3 == 1 ? mm : mr);
                const float4 wr = p.wd[i * p.C4 + c4];
                wk[i] = make_float4(wr.x * m, wr.y * m, wr.z * m, wr.w * m);
            }
            const float4 dsc = p.dscale[c4], dsh = p.dshift[c4];
            struct Row { float4 l, m, r; float k; };
            auto load_row = [&](int ih) {
                Row q;
                const int ihc = min(max(ih, 0), p.H - 1);
                const float4* row = xin + (size_t)ihc * p.W * p.C4;
                q.l = row[cl]; q.m = row[cm]; q.r = row[cr];
                q.k = (ih >= 0 && ih < p.H) ? 1.f : 0.f;
                return q;
            };
            auto row_sum = [&](const Row& q, int b) {
                float4 a = make_float4(q.l.x * wk[b].x, q.l.y * wk[b].y, q.l.z * wk[b].z, q.l.w * wk[b].w);
                a = fma4(q.m, wk[b + 1], a);
                return fma4(q.r, wk[b + 2], a);
            };
            f16x4 ohi[RPT], olo[RPT];
            auto emit = [&](int j, const Row& a, const Row& b, const Row& c) {
                const float4 sa = row_sum(a, 0), sb = row_sum(b, 3), sc = row_sum(c, 6);
                float4 s = make_float4(sa.x * a.k, sa.y * a.k, sa.z * a.k, sa.w * a.k);
                s = make_float4(fmaf(sb.x, b.k, s.x), fmaf(sb.y, b.k, s.y), fmaf(sb.z, b.k, s.z), fmaf(sb.w, b.k, s.w));
                s = make_float4(fmaf(sc.x, c.k, s.x), fmaf(sc.y, c.k, s.y), fmaf(sc.z, c.k, s.z), fmaf(sc.w, c.k, s.w));
                const float4 o = fma4(s, dsc, dsh);
                f32x4 v;
                v[0] = relu6(o.x); v[1] = relu6(o.y); v[2] = relu6(o.z); v[3] = relu6(o.w);
                v = v * p.a_scale;
                ohi[j] = __builtin_convertvector(v, f16x4);
                olo[j] = __builtin_convertvector(v - __builtin_convertvector(ohi[j], f32x4), f16x4);
            };
            if (STRIDE == 1) {
                Row r0 = load_row(ih0), r1 = load_row(ih0 + 1), r2 = load_row(ih0 + 2), r3 = load_row(ih0 + 3);
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    const Row r4 = load_row(ih0 + j + 4);
                    emit(j, r0, r1, r2);
                    r0 = r1; r1 = r2; r2 = r3; r3 = r4;
                }
            } else {
                Row r0 = load_row(ih0), r1 = load_row(ih0 + 1), r2 = load_row(ih0 + 2);
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    const Row n1 = load_row(ih0 + 2 * j + 3), n2 = load_row(ih0 + 2 * j + 4);
                    emit(j, r0, r1, r2);
                    r0 = r2; r1 = n1; r2 = n2;
                }
            }
            // ---- publish ----
            __syncthreads();   // every wave is done reading the previous chunk's tiles
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int R = (row0 + j) * TW + tw;
                *(f16x4*)(&As[swzb(R, c4l >> 1) + 8 * (c4l & 1)]) = ohi[j];
                *(f16x4*)(&As[swzb(R, 4 + (c4l >> 1)) + 8 * (c4l & 1)]) = olo[j];
            }
#pragma unroll
            for (int q = 0; q < BP; ++q) *(f32x4*)(&Bs[swzb(srow + 32 * q, skq)]) = rb[q];
            __syncthreads();
            // ---- contract ----
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 ah[2], al[2], bh[NI], bl[NI];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    ah[mi] = *(const f16x8*)(&As[swzb(arow + mi * 32, 2 * s + lh)]);
                    al[mi] = *(const f16x8*)(&As[swzb(arow + mi * 32, 4 + 2 * s + lh)]);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    bh[ni] = *(const f16x8*)(&Bs[swzb(brow + ni * 32, 2 * s + lh)]);
                    bl[ni] = *(const f16x8*)(&Bs[swzb(brow + ni * 32, 4 + 2 * s + lh)]);
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    }
            }
        }
        // ---- epilogue: tile row R = th*TW + tw; accumulator register r of lane-half lh -> R = base + (r&3) + 8*(r>>2) + 4*lh
        // (full patches store unconditionally: a per-store bounds branch costs an s_waitcnt vmcnt(0) per store)
        const bool full = oh0 + TH <= p.OH && ow0 + TW <= p.OW;
        float* ybase = p.y + ((size_t)n * p.OH * p.OW) * p.Cout;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = n0 + wn * WN + ni * 32 + li;
            const float ds = p.descale[col], sh = p.pshift[col];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int Rb = wm * 64 + mi * 32 + 4 * lh;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int R = Rb + (r & 3) + 8 * (r >> 2);
                        ybase[((size_t)(oh0 + R / TW) * p.OW + ow0 + (R % TW)) * p.Cout + col] = apply_act<ACT>(fmaf(acc[mi][ni][r], ds, sh));
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int R = Rb + (r & 3) + 8 * (r >> 2);
                        const int oh = oh0 + R / TW, owc = ow0 + (R % TW);
                        if (oh < p.OH && owc < p.OW)
                            ybase[((size_t)oh * p.OW + owc) * p.Cout + col] = apply_act<ACT>(fmaf(acc[mi][ni][r], ds, sh));
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA form (stride 1, up to 256 channels): ONE persistent workgroup of 8 waves per CU, everything the block needs
// staged by LDS-DMA.
//
// The general kernel above keeps each chunk's 18 input float4 per thread in registers and consumes them at once: every
// chunk pays a memory round trip, and it lost to the two-kernel form for C >= 128.  Here nothing the depthwise reads
// passes through a VGPR on its way in (an earlier generation with all 8 waves alternating depthwise and MFMA phases --
// "v2", two barriers per chunk -- was retired in round 2; git history has it):
//   * per 32-channel chunk the (TH+2) x (TW+2) halo patch (180 pixels x 128 B) and the chunk's slice of the split weight
//     rows (BN x 128 B) go global -> LDS with `buffer_load_dwordx4 ... lds`, ONE CHUNK AHEAD of their use (two stages);
//     the only thing that changes from chunk to chunk is the SGPR offset.  Pixels outside the image are given an
//     out-of-range offset: the buffer bounds check writes ZEROS into LDS for them, which is the convolution's padding --
//     no masks, no clamps in the arithmetic.  The weight rows use the swizzled-source / linear-destination form of
//     pwconv_f32.hip so that the fragment reads are the conflict-free ones of pwconv_f16s.hip;
//   * the depthwise taps, scale and shift for all C channels sit in LDS for the life of the workgroup;
//   * depthwise: thread = (channel quad, column, 2 output rows), 12 ds_read_b128 (a wave reads 8 pixels x 128 B = 1 KB
//     contiguous) -> ReLU6 -> x 2^12 -> f16 hi/lo -> A tile;
//   * contraction as in pwconv_f16s.hip (operands swapped, 32x32x16 f16 MFMA, 3 products), waves as 2 (M) x 4 (N);
//   * epilogue through a wave-private 4 KB scratch in the stage the last chunk has just released: 128-B-line stores whose
//     per-lane offsets are out of range for pixels beyond the map (partial patches cost no branches).
// The (patch, chunk) steps of a workgroup form one flat sequence, so the first chunk of the next patch streams in under
// the last MFMAs and the stores of the current one.
constexpr int HALO_B = 192 * 128;   // 180 pixels used, 24 wave-instructions of 8 pixels
constexpr int CMAX2 = 512;          // channel bound of the resident depthwise constants (11 floats per channel)

// ---------------------------------------------------------------------------------------------------------------------
// The waves are SPECIALISED.  With all 8 waves running the depthwise phase, meeting at a barrier, running the MFMA phase
// and meeting again, the vector ALU idles during the contraction and the matrix pipe during the depthwise, and
// in-kernel stamps put 45 % of a wave's life in barriers.  Here waves 0-3 (one per SIMD) are PRODUCERS -- halo DMA and the
// depthwise of step g+1 into A[(g+1) & 1] -- while waves 4-7 are CONSUMERS -- weight DMA, the MFMAs of step g from
// A[g & 1], and the epilogue.  One barrier per step; on every SIMD a producer's VALU/LDS work runs beside a consumer's
// matrix work.  BN = 256: EIGHT consumer waves (4-11, two per SIMD) of 64 x 64 each instead of four of 64 x 128 -- the consumers were
// the patch's critical path (16.4 k cycles of MFMA + 8.4 k of weight-piece issue + 7.2 k of epilogue stores against the producers'
// 19.8 k of depthwise, profiles/r05_blk_stamps.txt): the matrix work per SIMD is the same, but the epilogue and the weight DMA of a
// patch are now split eight ways, and a wave holds 64 accumulators instead of 128 (twelve waves per CU: 168 registers each).  The halo ring is HS deep (3 where LDS allows: two steps of lookahead for the HBM stream; vmcnt retires
// in order, so a producer waits with one step's pieces still in flight).  A consumer's epilogue scratch is the part of
// the weight stage its own next DMA pieces will overwrite, so it needs no synchronisation beyond program order.
template <int TW, int BN, int HS, int ACT>
__global__ __launch_bounds__(256 + 64 * (BN == 256 ? 8 : 4), BN == 256 ? 1 : 2) void dwpw3_f16s_kernel(DwPwSParams p) {
    constexpr int NCW = BN == 256 ? 8 : 4;                      // consumer waves: 2 (M) x NCW / 2 (N)
    constexpr int CWN = NCW / 2;
    constexpr int NTHR = 256 + 64 * NCW;
    constexpr int TH = 128 / TW, HC = TW + 2;
    constexpr int WCAP = 256;                                   // channel capacity of the resident depthwise constants
    constexpr int B_ST = BN * ROWB;
    constexpr int B_OFF = HS * HALO_B, A_OFF = B_OFF + 2 * B_ST, W_OFF = A_OFF + 2 * 128 * ROWB, E_OFF = W_OFF + 11 * WCAP * 4;
    constexpr int WN = BN / CWN, NI = WN / 32;                  // consumer wave tile 64 x 64
    constexpr int BPW = BN / 8 / NCW;                           // weight DMA pieces per consumer wave (its scratch: BPW KB = 4)
    static_assert(BPW == 4 && NI == 2, "a consumer's slice of the weight stage is its 4 KB epilogue scratch");
    constexpr bool HALO_BY_CONSUMER = false;                    // (see the halo DMA duty note below)
    constexpr int HPW = 24 / (HALO_BY_CONSUMER ? NCW : 4);       // halo DMA pieces per issuing wave
    static_assert(E_OFF + 2 * CMAX2 * 4 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[E_OFF + 2 * CMAX2 * 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C4 * 4, KT = p.KT;
    {
        float4* wl = (float4*)(smem + W_OFF);
        for (int i = tid; i < 9 * p.C4; i += NTHR) wl[i] = p.wd[i];
        for (int i = tid; i < p.C4; i += NTHR) { wl[9 * p.C4 + i] = p.dscale[i]; wl[10 * p.C4 + i] = p.dshift[i]; }
        float* el = (float*)(smem + E_OFF);
        for (int i = tid; i < p.Cout; i += NTHR) { el[i] = p.descale[i]; el[CMAX2 + i] = p.pshift[i]; }
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (long long)p.nimg * p.H * p.W * C * 4);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.wsplit, (long long)p.Cout * C * 4);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y, (long long)p.nimg * p.OH * p.OW * p.Cout * 4);
    const unsigned OOB = 0xFFFFFFF0u;
    const unsigned nitem = (p.total - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int nsteps = (int)(nitem * KT);
    struct Item { int n, oh0, ow0, n0; };
    auto decode = [&](unsigned i) {
        const unsigned lt = xcd_remap_dir(blockIdx.x + (i < nitem ? i : nitem - 1) * gridDim.x, p.total, p.reverse);
        Item it;                                       // three multiply-high quotients instead of five 32-bit divisions
        const unsigned pt = fastdiv(lt, p.m_n);
        it.n0 = (lt - pt * p.tiles_n) * BN;
        const unsigned pr = fastdiv(pt, p.m_w);         // patch row index over all images
        it.ow0 = (pt - pr * p.tiles_w) * TW;
        it.n = fastdiv(pr, p.m_h);
        it.oh0 = (pr - it.n * p.tiles_h) * TH;
        return it;
    };
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    auto piece = [&](const __amdgpu_buffer_rsrc_t& r, unsigned lds_addr, unsigned voff, unsigned soff) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff),
                     "s"(r), "s"(__builtin_amdgcn_readfirstlane(soff))
                     : "memory", "m0");   // issued from asm so that hipcc does not drain vmcnt before every LDS read
    };
    // halo DMA duty: producers.  (Measured both ways with in-kernel stamps: the patch time does not change, the waiting moves to
    // whichever waves issue the vector-memory instructions.  Neither did keeping all the weight rows of the 128-channel block
    // resident in LDS (-29 % instructions) nor slicing the epilogue's stores over the next patch's steps: at 611 MB per
    // launch in ~145 us the block runs at 4.2 TB/s of HBM traffic, 85 % of what a streaming copy reaches on the same box.)
    // (Round 5, BN = 256 with eight consumer waves: the producers are then the patch's critical path by the stamps -- 18.9 k cycles of
    // depthwise + 5.2 k of halo issue against 6.7 k of MFMA + 7.4 k of epilogue per consumer -- and the halo duty on the consumers, three
    // pieces each, was still SLOWER: 99 -> 108 us.  With a two-deep halo ring the issuing wave waits for its pieces inside the step.)
    const int hw = HALO_BY_CONSUMER ? wave - 4 : wave & 3;      // index of this wave among those that issue the halo pieces
    unsigned hv[HPW];
    unsigned pf_i = 0;
    int pf_kc = 0, pf_step = 0;
    auto setup_halo = [&](unsigned i) {
        const Item it = decode(i);
#pragma unroll
        for (int j = 0; j < HPW; ++j) {
            const int q = (hw * HPW + j) * 8 + (lane >> 3);
            const int hr = q / HC, hc = q - hr * HC;
            const int ih = it.oh0 - p.pad_t + hr, iw = it.ow0 - p.pad_l + hc;
            const bool ok = q < (TH + 2) * HC && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            hv[j] = ok ? ((unsigned)((it.n * p.H + ih) * p.W + iw) * (unsigned)C + 4u * (lane & 7)) * 4u : OOB;
        }
    };
    auto halo_dma = [&]() {          // pieces of step pf_step into ring slot pf_step % HS
        const unsigned base = lds0 + (pf_step % HS) * HALO_B;
        const unsigned so = (unsigned)pf_kc * 128u;
#pragma unroll
        for (int j = 0; j < HPW; ++j) piece(rx, base + (hw * HPW + j) * 1024, hv[j], so);
        ++pf_step;
        if (++pf_kc == KT) {
            pf_kc = 0;
            setup_halo(++pf_i);
        }
    };
    STEM_STAMP_DECL;

    if (wave < 4) {
        // =============================== producer: halo DMA + depthwise ===============================
        const int quad = tid & 7, col = (tid >> 3) % TW, r0 = 4 * (tid / (8 * TW));
        const unsigned char* hsrc0 = smem + ((r0 * HC + col) * 128 + quad * 16);
        if (!HALO_BY_CONSUMER) {
            setup_halo(0);
#pragma unroll
            for (int i = 0; i < HS - 1; ++i)
                if (i < nsteps) halo_dma();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int dkc = 0;                      // chunk of the step the depthwise works on (step g + 1)
        for (int g = -1; g < nsteps; ++g) {
            const bool issued = !HALO_BY_CONSUMER && pf_step < nsteps;
            if (issued) halo_dma();
            STEM_STAMP(1);
            if (g + 1 < nsteps) {
                // f32x4 vector arithmetic: each operation becomes two v_pk_*_f32 on the register pairs the ds_read_b128
                // delivered (written on float4 structs the SLP vectoriser paired lanes of DIFFERENT rows: four v_mov per
                // packed op).  Same IEEE operations per element, same order as dwconv.hip: bit-identical to it.
                const f32x4* wl = (const f32x4*)(smem + W_OFF) + dkc * 8 + quad;
                f32x4 wk[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) wk[i] = wl[i * p.C4];
                const f32x4 dsc = wl[9 * p.C4], dsh = wl[10 * p.C4];
                const unsigned char* hs = hsrc0 + ((g + 1) % HS) * HALO_B;
                unsigned char* At = smem + A_OFF + ((g + 1) & 1) * (128 * ROWB);
                f32x4 h[6][3];
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) h[a][b] = *(const f32x4*)(hs + (a * HC + b) * 128);
                auto row_sum = [&](int a, int b) {
                    f32x4 t = h[a][0] * wk[b];
                    t = __builtin_elementwise_fma(h[a][1], wk[b + 1], t);
                    return __builtin_elementwise_fma(h[a][2], wk[b + 2], t);
                };
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 sa = row_sum(j, 0), sb = row_sum(j + 1, 3), sc = row_sum(j + 2, 6);
                    const f32x4 sacc = (sa + sb) + sc;
                    const f32x4 o = __builtin_elementwise_fma(sacc, dsc, dsh);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = relu6(o[e]);
                    v = v * p.a_scale;
                    const f16x4 hi = __builtin_convertvector(v, f16x4);
                    const f16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), f16x4);
                    const int R = (r0 + j) * TW + col;
                    *(f16x4*)(At + swzb(R, quad >> 1) + 8 * (quad & 1)) = hi;
                    *(f16x4*)(At + swzb(R, 4 + (quad >> 1)) + 8 * (quad & 1)) = lo;
                }
                if (++dkc == KT) { dkc = 0; STEM_STAMP_COUNT; }
            }
            STEM_STAMP(2);
            // the halo of step g + 2 must have landed before the next iteration reads it; with a 3-deep ring the pieces
            // issued in this iteration (step g + 3) stay in flight
            if (HS >= 3 && issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STEM_STAMP(0);
            __syncthreads();
            STEM_STAMP(3);
        }
        STEM_STAMP_FLUSH(p.stamps, lane, wave);
    } else {
        // =============================== consumer: weight DMA + MFMA + epilogue ===============================
        const int cw = wave - 4, wm = cw / CWN, wn = cw % CWN;
        const int li = lane & 31, lh = lane >> 5;
        const int arow = wm * 64 + li, brow = wn * WN + li;
        const int erow = lane >> 3, ech = lane & 7;
        unsigned bv[BPW];
        unsigned pb_i = 0;
        int pb_kc = 0, pb_step = 0;
        auto setup_b = [&](unsigned i) {
            const Item it = decode(i);
#pragma unroll
            for (int j = 0; j < BPW; ++j) {
                const int r = (cw * BPW + j) * 8 + (lane >> 3);
                bv[j] = ((unsigned)(it.n0 + r) * (unsigned)C + 4u * ((lane & 7) ^ ((r >> 1) & 7) ^ ((r & 1) << 2))) * 4u;
            }
        };
        auto b_dma = [&]() {
            const unsigned base = lds0 + B_OFF + (pb_step & 1) * B_ST;
            const unsigned so = (unsigned)pb_kc * 128u;
#pragma unroll
            for (int j = 0; j < BPW; ++j) piece(rw, base + (cw * BPW + j) * 1024, bv[j], so);
            ++pb_step;
            if (++pb_kc == KT) {
                pb_kc = 0;
                setup_b(++pb_i);
            }
        };
        f32x16 acc[2][NI];
        auto zero_acc = [&]() {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        };
        auto epilogue = [&](const Item& cur, int stage) {
            // lane (li, lh) holds row m = li, columns 4*lh + 8*(r >> 2) + (r & 3) of each 32 x 32 block (operands swapped)
            unsigned char* scr = smem + B_OFF + stage * B_ST + cw * (BPW * 1024);
            const float* el = (const float*)(smem + E_OFF);
            unsigned sv[2][4];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int R = wm * 64 + mi * 32 + erow + 8 * i;
                    const int oh = cur.oh0 + R / TW, ow = cur.ow0 + R % TW;
                    sv[mi][i] = (oh < p.OH && ow < p.OW)
                                    ? ((unsigned)((cur.n * p.OH + oh) * p.OW + ow) * (unsigned)p.Cout + (unsigned)(cur.n0 + wn * WN + 4 * ech)) * 4u
                                    : OOB;
                }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int nl = cur.n0 + wn * WN + ni * 32 + 4 * ech;
                const f32x4 ds = *(const f32x4*)(el + nl), sh = *(const f32x4*)(el + CMAX2 + nl);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * j + e];
                        *(f32x4*)(scr + li * 128 + 16 * ((2 * j + lh) ^ (li & 7))) = v;
                    }
                    // (all four reads first: each store below is an asm statement that clobbers memory, and with a read per store
                    // hipcc waited out one LDS round trip per 16 bytes -- 64 of them per patch and wave at BN = 256.  Going further --
                    // parking and fetching block k + 1 before block k is stored -- measured no better)
                    f32x4 rv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = erow + 8 * i;
                        rv[i] = *(const f32x4*)(scr + r * 128 + 16 * (ech ^ (r & 7)));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 v = rv[i];
                        const f32x4 z = __builtin_elementwise_fma(v, ds, sh);
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = apply_act<ACT>(z[e]);
                        bstore16_welded(o, ry, sv[mi][i], __builtin_amdgcn_readfirstlane((unsigned)(ni * 32) * 4u));
                    }
                }
            }
        };
        zero_acc();
        setup_b(0);
        Item cur = decode(0);
        unsigned ci = 0;
        int ckc = 0;
        bool pending = false;             // the previous step closed a patch: its accumulators wait for their epilogue
        if (HALO_BY_CONSUMER) {
            setup_halo(0);
#pragma unroll
            for (int i = 0; i < HS - 1; ++i)
                if (i < nsteps) halo_dma();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                  // (the producers' prologue barrier)
        for (int g = -1; g < nsteps; ++g) {
            if (pending) {
                epilogue(cur, (g + 1) & 1);          // scratch: the slice of weight stage (g + 1) & 1 this wave refills next
                zero_acc();
                cur = decode(++ci);
                pending = false;
                STEM_STAMP_COUNT;
            }
            STEM_STAMP(5);
            if (pb_step < nsteps) b_dma();             // weights of step g + 1 into stage (g + 1) & 1
            const bool halo_issued = HALO_BY_CONSUMER && pf_step < nsteps;
            if (halo_issued) halo_dma();               // halo of step g + HS (issued AFTER the weights: vmcnt retires in order)
            STEM_STAMP(1);
            if (g >= 0) {
                const unsigned char* As = smem + A_OFF + (g & 1) * (128 * ROWB);
                const unsigned char* Bs = smem + B_OFF + (g & 1) * B_ST;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f16x8 ah[2], al[2], bh[NI], bl[NI];
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        ah[mi] = *(const f16x8*)(As + swzb(arow + mi * 32, 2 * s + lh));
                        al[mi] = *(const f16x8*)(As + swzb(arow + mi * 32, 4 + 2 * s + lh));
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        bh[ni] = *(const f16x8*)(Bs + swzb(brow + ni * 32, 2 * s + lh));
                        bl[ni] = *(const f16x8*)(Bs + swzb(brow + ni * 32, 4 + 2 * s + lh));
                    }
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ni], al[mi], acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ni], ah[mi], acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ni], ah[mi], acc[mi][ni], 0, 0, 0);
                        }
                }
                if (++ckc == KT) { ckc = 0; pending = true; }
            }
            STEM_STAMP(4);
            // the next step's weights, the halo of step g + 2 (and this patch's stores) have landed; with a 3-deep ring
            // the halo pieces issued in this iteration stay in flight
            if (HS >= 3 && halo_issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STEM_STAMP(0);
            __syncthreads();
            STEM_STAMP(3);
        }
        if (pending) { epilogue(cur, (nsteps - 1) & 1); STEM_STAMP_COUNT; }
#ifdef HSEFR_STEM_STAMPS
        if (lane == 0 && p.stamps && blockIdx.x < 256 && cw < 4) {      // consumer waves (the first four) report in the upper half of the stamp table
            unsigned long long* o = p.stamps + ((blockIdx.x + 256) * 4 + cw) * 10;
            for (int i_ = 0; i_ < 8; ++i_) o[i_] = st[i_];
            o[8] = __builtin_amdgcn_s_memtime() - tstart;
            o[9] = npatch;
        }
#endif
    }
}

HSEFR_KNOB(g_tw, 0);   // dev builds: 0 = auto, 8 | 16 = forced patch width
HSEFR_KNOB(g_bn, 0);   // dev builds: 0 = auto, 64 | 128 | 256 = forced N tile

template <int STRIDE, int TW, int BN, int OCC>
int launch_t(DwPwSParams& p, int n, int act, hipStream_t s) {
    constexpr int TH = 128 / TW;
    p.tiles_w = (p.OW + TW - 1) / TW;
    p.tiles_h = (p.OH + TH - 1) / TH;
    p.tiles_n = p.Cout / BN;
    const long long total = (long long)n * p.tiles_w * p.tiles_h * p.tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "dwpw_f16split: grid too large");
    p.total = (unsigned)total;
    const unsigned cap = 256u * OCC;
    const unsigned g = p.total < cap ? p.total : cap;
#define HSEFR_DWPWS(A) HSEFR_LAUNCH((dwpw_f16s_kernel<STRIDE, TW, BN, OCC, A>), dim3(g), dim3(256), 0, s, p)
    if (act == HSEFR_ACT_RELU6) HSEFR_DWPWS(HSEFR_ACT_RELU6);
    else if (act == HSEFR_ACT_RELU) HSEFR_DWPWS(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_NONE) HSEFR_DWPWS(HSEFR_ACT_NONE);
    else { set_error("dwpw_f16split: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_DWPWS
    return launch_status("dwpw_f16split");
}

template <int STRIDE, int TW>
int launch_bn(DwPwSParams& p, int n, int bn, int act, hipStream_t s) {
    if (bn == 64) return launch_t<STRIDE, TW, 64, 2>(p, n, act, s);
#ifdef HSEFR_DEV
    // (the 256-column tile spills: only the "dwpws_bn" knob of development builds asks for it -- twelve instantiations the product does not carry)
    if (bn == 256) return launch_t<STRIDE, TW, 256, 2>(p, n, act, s);
#endif
    return launch_t<STRIDE, TW, 128, 2>(p, n, act, s);
}

template <int TW, int BN, int HS>
int launch_v3(DwPwSParams& p, int n, int act, hipStream_t s) {
    constexpr int TH = 128 / TW;
    p.tiles_w = (p.OW + TW - 1) / TW;
    p.tiles_h = (p.OH + TH - 1) / TH;
    p.tiles_n = p.Cout / BN;
    const long long total = (long long)n * p.tiles_w * p.tiles_h * p.tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31) / 16, HSEFR_ERR_UNSUPPORTED, "dwpw_f16split: grid too large");
    p.total = (unsigned)total;
    auto magic = [](unsigned d) { return d == 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); };
    p.m_n = magic(p.tiles_n); p.m_w = magic(p.tiles_w); p.m_h = magic(p.tiles_h);
    HSEFR_REQUIRE(total * (long long)(p.tiles_n > p.tiles_w ? (p.tiles_n > p.tiles_h ? p.tiles_n : p.tiles_h) : (p.tiles_w > p.tiles_h ? p.tiles_w : p.tiles_h)) < (1ll << 32),
                  HSEFR_ERR_UNSUPPORTED, "dwpw_f16split: grid too large for the quotient multipliers");
    const unsigned g = p.total < 256u ? p.total : 256u;
#define HSEFR_DWPW3(A) HSEFR_LAUNCH((dwpw3_f16s_kernel<TW, BN, HS, A>), dim3(g), dim3(256 + 64 * (BN == 256 ? 8 : 4)), 0, s, p)
    if (act == HSEFR_ACT_RELU6) HSEFR_DWPW3(HSEFR_ACT_RELU6);
    else if (act == HSEFR_ACT_RELU) HSEFR_DWPW3(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_NONE) HSEFR_DWPW3(HSEFR_ACT_NONE);
    else { set_error("dwpw_f16split: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_DWPW3
    return launch_status("dwpw_f16split");
}

}  // namespace

#ifdef HSEFR_DEV
void set_dwpws_tw(int v) { g_tw = v; }
void set_dwpws_bn(int v) { g_bn = v; }
#endif

bool dwpw_f16s_supported(int c, int cout, int stride) {
    return c > 0 && c % 32 == 0 && cout > 0 && cout % 64 == 0 && (stride == 1 || stride == 2);
}

int launch_dwpw_f16s(const float* x, const float* wd, const float* dscale, const float* dshift, const void* wsplit,
                     const float* descale, const float* pshift, float* y, int n, int h, int w, int c, int stride, int pad_t,
                     int pad_l, int oh, int ow, int cout, int a_log2, int act, hipStream_t s) {
    HSEFR_REQUIRE(dwpw_f16s_supported(c, cout, stride), HSEFR_ERR_UNSUPPORTED,
                  "dwpw_f16split: c=%d cout=%d stride=%d not covered (c %% 32, cout %% 64, stride 1|2)", c, cout, stride);
    HSEFR_REQUIRE(n >= 0 && h > 0 && w > 0 && oh > 0 && ow > 0, HSEFR_ERR_INVALID, "dwpw_f16split: bad shape");
    HSEFR_REQUIRE(a_log2 > 0 && a_log2 <= 12, HSEFR_ERR_INVALID, "dwpw_f16split: a_log2=%d (the depthwise result is in [0,6]: 1..12)", a_log2);
    if (n == 0) return HSEFR_OK;
    DwPwSParams p;
    p.x = (const float4*)x; p.wd = (const float4*)wd; p.dscale = (const float4*)dscale; p.dshift = (const float4*)dshift;
    p.wsplit = (const float*)wsplit; p.descale = descale; p.pshift = pshift; p.y = y;
    p.H = h; p.W = w; p.C4 = c / 4; p.KT = c / 32; p.OH = oh; p.OW = ow; p.Cout = cout; p.pad_t = pad_t; p.pad_l = pad_l;
    p.a_scale = ldexpf(1.f, a_log2);
    p.reverse = sweep_reverse();
    p.nimg = n;
    p.stamps = nullptr;
#ifdef HSEFR_STEM_STAMPS
    p.stamps = stamp_buffer(s);
#endif
    const long long in_bytes = (long long)n * h * w * c * 4, out_bytes = (long long)n * oh * ow * cout * 4;
    // wave-specialised LDS-DMA form (resident depthwise constants for up to 256 channels); everything else -- stride 2,
    // more than 256 channels, cout % 128 != 0, tensors beyond 4 GB -- takes the general K-chunked kernel below
    if (stride == 1 && c <= 256 && cout <= CMAX2 && cout % 128 == 0 && in_bytes < (1ll << 32) - 16 && out_bytes < (1ll << 32) - 16) {
        // patch shape: 8 x 16 or 16 x 8 output pixels, whichever wastes fewer (partial patches cost compute, not bytes)
        auto padded = [&](int tw) { const int th = 128 / tw; return (long long)((ow + tw - 1) / tw * tw) * ((oh + th - 1) / th * th); };
        int tw = padded(16) <= padded(8) ? 16 : 8;
        if (g_tw == 8 || g_tw == 16) tw = g_tw;
        int bn = cout % 256 == 0 ? 256 : 128;
        if ((g_bn == 128 || g_bn == 256) && cout % g_bn == 0) bn = g_bn;
        if (bn == 256) return tw == 16 ? launch_v3<16, 256, 2>(p, n, act, s) : launch_v3<8, 256, 2>(p, n, act, s);
        return tw == 16 ? launch_v3<16, 128, 3>(p, n, act, s) : launch_v3<8, 128, 3>(p, n, act, s);
    }
    // N tile: 128 output channels (64 accumulator registers; the 256 variant spills); wider layers redo the depthwise
    // work once per N tile, the tiles of one patch running side by side on one XCD so that the re-read hits its L2
    int bn = cout % 128 == 0 ? 128 : 64;
    if (g_bn && cout % g_bn == 0) bn = g_bn;
    // patch shape: 8 rows x 16 columns unless the width only tiles by 8
    int tw = (ow % 16 == 0 || ow % 8 != 0) ? 16 : 8;
    if (ow <= 8) tw = 8;
    if (g_tw == 8 || g_tw == 16) tw = g_tw;
    if (stride == 1) return tw == 16 ? launch_bn<1, 16>(p, n, bn, act, s) : launch_bn<1, 8>(p, n, bn, act, s);
    return tw == 16 ? launch_bn<2, 16>(p, n, bn, act, s) : launch_bn<2, 8>(p, n, bn, act, s);
}

}  // namespace hsefr
