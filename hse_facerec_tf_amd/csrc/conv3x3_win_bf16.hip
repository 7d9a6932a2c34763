// (Round 5: superseded on every ResNet-50 shape by conv3x3_w2_bf16.hip -- four wide MFMA waves, padded-row pixel mapping, a swizzle that is
// conflict-free under every tap; this kernel's stamps are what showed why: profiles/r05_conv_stamps.txt.  It still serves 3x3 layers the
// new one's geometries do not prefer -- maps 40-47 pixels wide -- and stays the A/B reference of tools/kbench_conv.py.)
// 3x3 / stride 1 / pad 1 bf16 convolution + scale + shift (+ residual) + act with the INPUT WINDOW resident in LDS: the sixteen
// 3x3 layers of ResNet-50's bottlenecks (resnet50_ft, the graph behind vgg2_resnet.pb at facerec_test.py:213).  NHWC bf16 in / out,
// fp32 accumulation, gfx950.  Same results contract as conv_bf16.hip / conv_dma_bf16.hip (rounding points of oracle/resnet50.py).
//
// conv_dma_bf16.hip gathers the im2col tile tap by tap: every input pixel crosses L2 -> LDS nine times, and its four loader waves
// (one 1-KiB LDS-DMA piece per ~150 cycles each) set the pace -- the MFMA waves sat at the step barrier 40 % of the time (stamps).
// A 3x3 / stride-1 tap is only a SHIFT of the same pixels, so here
//   * a tile is TR full-width output rows of IMG images; its (TR + 2) x (W + 2) input window of one 64-channel slab goes global -> LDS
//     ONCE (LDS-DMA, out-of-image pixels zero-filled by the buffer bounds check: the padding costs nothing), double-buffered: the
//     next slab's window arrives during the nine tap-steps of the current one;
//   * window rows are 128-B LDS rows at a pitch of PITCH = 16 k >= W + 2 rows per image row, with the usual chunk swizzle keyed on
//     the row index.  A tap (kh, kw) moves a pixel's row by kh * PITCH + kw: the key depends on kw only, so three pre-swizzled
//     offsets per row block serve all nine taps and a fragment address is one add;
//   * per tap-step only the WEIGHT tile (BN x 64 of one tap and slab) is streamed (3-stage ring): 16 KiB instead of 45 KiB per step
//     -- the loaders are no longer the bottleneck, the K loop is MFMA-bound;
//   * everything else is conv_dma_bf16.hip: 8 MFMA waves (v_mfma_f32_16x16x32_bf16, weights first, permuted weight rows so a
//     lane owns 8 consecutive channels) + 4 loader waves, one barrier per step, epilogue straight from the accumulators.
// Tiles: 14x14 maps: one image (196 pixels, 13 of 14 row blocks used) x 128 channels; 28x28: 7 rows; 56x56: 4 rows x 64 channels;
// 7x7: two images.  Every output element is accumulated in one fixed order by one wave: bit-identical run to run.
#include <type_traits>

#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ROWB = 128;                 // bytes per LDS row: 64 bf16

#ifdef HSEFR_CD_STAMPS
__device__ unsigned long long g_w3_stamps[256 * 12 * 8];
#define W3_STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); st[i] += _t - tprev; tprev = _t; } while (0)
#define W3_STAMP_DECL unsigned long long st[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tprev = __builtin_amdgcn_s_memtime(); const unsigned long long tstart = tprev
#define W3_STAMP_FLUSH do { if (lane == 0 && blockIdx.x < 256) { unsigned long long* o = g_w3_stamps + (blockIdx.x * 12 + wave) * 8; \
    for (int i_ = 0; i_ < 6; ++i_) o[i_] = st[i_]; o[6] = __builtin_amdgcn_s_memtime() - tstart; o[7] = nsteps; } } while (0)
#else
#define W3_STAMP(i) do { } while (0)
#define W3_STAMP_DECL do { } while (0)
#define W3_STAMP_FLUSH do { } while (0)
#endif
constexpr int WIN_BYTES = 384 * ROWB;     // one window buffer: up to 384 rows (48 KiB)
constexpr int WSLOTS = 12;                // window pieces per loader wave (4 x 12 x 8 rows = 384)

__device__ __forceinline__ int swz_key(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 2); }
__device__ __forceinline__ unsigned f2bf_bits(float f) { return hsefr_bf16_bits(f); }      // round-to-nearest-even (common.h)
__device__ __forceinline__ float bfround(float f) { return __uint_as_float(f2bf_bits(f) << 16); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_sgpr(const void* ptr, long long bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned n = __builtin_amdgcn_readfirstlane(bytes <= 0 ? 0u : (bytes > 0xffffffffll ? 0xffffffffu : (unsigned)bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}

struct Win3Params {
    const void* x;       // [N,H,W,C] bf16
    const void* wt;      // [Cout][9*C] bf16, k = (kh*3 + kw)*C + c
    const float* scale;
    const float* shift;
    const void* res;     // [M,Cout] bf16 or null
    void* y;             // [M,Cout] bf16
    long long x_bytes;
    int N, H, W, C, Cout;
    int TR, IMG, PITCH;  // tile = IMG images x TR rows x W columns (IMG > 1 only with TR == H); window pitch in rows
    int tiles_y;         // H / TR
    float act_lo, act_hi;
    unsigned M;
    unsigned tiles_n, total_tiles;
    int reverse;
};

template <int RBW, int WAVES_M>
__global__ __launch_bounds__(768, 1) void conv3x3_win_bf16_kernel(Win3Params p) {
    constexpr int WAVES_N = 8 / WAVES_M;
    constexpr int BN = WAVES_N * 32;
    constexpr int BPW = BN / 32;                    // weight pieces per loader wave and step
    constexpr int BSTAGE = BN * ROWB;
    constexpr int RING_OFF = 2 * WIN_BYTES;
    constexpr int E_OFF = RING_OFF + 3 * BSTAGE;
    constexpr int DUMMY_OFF = E_OFF + 4096;         // 1 KiB that absorbs the pieces issued only to keep the counts fixed
    static_assert(DUMMY_OFF + 1024 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[DUMMY_OFF + 1024];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int CS = p.C / 64;
    if (blockIdx.x >= p.total_tiles) return;
    const unsigned ntile = (p.total_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const unsigned nq = ntile * (unsigned)CS;       // slab-windows this workgroup goes through
    const unsigned nsteps = nq * 9u;
    const unsigned wrowbytes = (unsigned)(9 * CS) * 128u;
    const int NPIX = p.IMG * p.TR * p.W;
    const int WROWS_IMG = (p.TR + 2) * p.PITCH;     // window rows per image
    const int WP = p.IMG * WROWS_IMG / 8;           // window pieces

    // tile i of this workgroup -> first image, first row, first output pixel, first output channel
    auto tile_origin = [&](unsigned i, int& tn0, int& ty0, unsigned& mm0, int& cc0) __attribute__((always_inline)) {
        const unsigned lt = xcd_remap_dir(blockIdx.x + (i < ntile ? i : ntile - 1) * gridDim.x, p.total_tiles, p.reverse);
        const unsigned tm = lt / p.tiles_n;
        cc0 = (int)(lt - tm * p.tiles_n) * BN;
        const unsigned ng = tm / (unsigned)p.tiles_y;
        tn0 = (int)ng * p.IMG;
        ty0 = (int)(tm - ng * (unsigned)p.tiles_y) * p.TR;
        mm0 = ((unsigned)tn0 * (unsigned)p.H + (unsigned)ty0) * (unsigned)p.W;
    };
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

    if (wave >= 8) {
        // =================================== loader waves 8..11 ===================================
        const int lw = wave - 8;
        auto piece = [&](const __amdgpu_buffer_rsrc_t& r, unsigned lds_addr, unsigned voff) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(r)
                         : "memory", "m0");
        };
        const __amdgpu_buffer_rsrc_t rx = make_rsrc_sgpr(p.x, p.x_bytes);
        // weight pieces: rows lw * BPW * 8 .. of the tile's BN rows
        unsigned pvb[BPW];
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int r = (lw * BPW + j) * 8 + (lane >> 3);
            pvb[j] = (unsigned)r * wrowbytes + 16u * (unsigned)((lane & 7) ^ swz_key(r));
        }
        // window pieces of this wave: piece lw + 4 s, s = 0 .. WSLOTS - 1 (those >= WP do not exist)
        unsigned wbase[WSLOTS];
        auto setup_window = [&](unsigned i) __attribute__((always_inline)) {     // slab-0 offsets of tile i's window
            int tn0, ty0, cc0;
            unsigned mm0;
            tile_origin(i, tn0, ty0, mm0, cc0);
#pragma unroll
            for (int s = 0; s < WSLOTS; ++s) {
                const int pw = lw + 4 * s;
                const int w = pw * 8 + (lane >> 3);
                const int img = w / WROWS_IMG, rem = w - img * WROWS_IMG;
                const int wy = rem / p.PITCH, wx = rem - wy * p.PITCH;
                const int n = tn0 + img, iy = ty0 - 1 + wy, ix = wx - 1;
                const bool ok = pw < WP && n < p.N && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                wbase[s] = ok ? (unsigned)((n * p.H + iy) * p.W + ix) * (unsigned)(p.C * 2) + 16u * (unsigned)((lane & 7) ^ swz_key(w)) : 0x80000000u;
            }
        };
        // issue window slots [s0, s0 + cnt) of the window with sequence number wq (buffer wq & 1), slab offset wsl; slots that do not
        // exist go to the dummy KiB with an out-of-range source (zero fill): the number of pieces per step stays fixed
        auto issue_window = [&](int s0, auto CNT, unsigned wq, unsigned wsl, bool live) __attribute__((always_inline)) {
            constexpr int cnt = decltype(CNT)::value;      // (s0 is a constant after unrolling: wbase stays in registers)
            unsigned voff[cnt];
#pragma unroll
            for (int s = 0; s < cnt; ++s) voff[s] = live ? wbase[s0 + s] + wsl * 128u : 0x80000000u;
#pragma unroll
            for (int s = 0; s < cnt; ++s) asm volatile("" : "+v"(voff[s]));
#pragma unroll
            for (int s = 0; s < cnt; ++s) {
                const int pw = lw + 4 * (s0 + s);
                const unsigned dst = (live && pw < WP) ? lds0 + (wq & 1u) * WIN_BYTES + (unsigned)pw * 1024u : lds0 + DUMMY_OFF;
                piece(rx, dst, voff[s]);
            }
        };
        // weights of one step: tile's channel origin cc0, K offset (tap * CS + slab) * 128
        const char* w_ptr = nullptr;
        long long w_bytes = 0;
        auto setup_weights = [&](unsigned i) __attribute__((always_inline)) {
            int tn0, ty0, cc0;
            unsigned mm0;
            tile_origin(i, tn0, ty0, mm0, cc0);
            w_ptr = (const char*)p.wt + (long long)cc0 * wrowbytes;
            w_bytes = (long long)(p.Cout - cc0) * wrowbytes;
        };

        // ---- prologue: window 0 completely, then the steady state ----
        setup_window(0);
        issue_window(0, std::integral_constant<int, WSLOTS>{}, 0u, 0u, true);
        unsigned nw_tile = CS > 1 ? 0u : 1u, nw_slab = CS > 1 ? 1u : 0u;      // (tile, slab) of the NEXT window (sequence number q + 1)
        if (CS == 1) setup_window(1);
        setup_weights(0);
        unsigned pf_tile = 0, pf_slab = 0, pf_step = 0;       // issue cursor
        unsigned ci = 0;                                        // tile the MFMA waves are on (for the epilogue constants)
        unsigned ckt = 0;
        unsigned k = 0;
        W3_STAMP_DECL;
        for (unsigned q = 0; q <= nq; ++q) {
            const bool next_live = q + 1 < nq;                  // window q + 1 exists
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (k >= nsteps + 2) break;
                if (k >= 2 && ckt == 0 && lw == 3) {
                    int tn0, ty0, e_c0;
                    unsigned mm0;
                    tile_origin(ci, tn0, ty0, mm0, e_c0);
                    const __amdgpu_buffer_rsrc_t rd = make_rsrc_sgpr(p.scale + e_c0, (long long)(p.Cout - e_c0) * 4),
                                                 rs = make_rsrc_sgpr(p.shift + e_c0, (long long)(p.Cout - e_c0) * 4);
                    const unsigned eb = lds0 + E_OFF + (ci & 1u) * 2048u;
                    piece(rd, eb, lane < 32 ? 16u * lane : 0x80000000u);
                    piece(rs, eb + 1024, lane >= 32 ? 16u * (unsigned)(lane - 32) : 0x80000000u);
                }
                // window q + 1 during taps 2..7 (its buffer was last read by slab q - 1, whose final step retires at the barrier
                // before issue step 9 q + 2), two slots per step
                if (t >= 2 && t <= 7)
                    issue_window(2 * (t - 2), std::integral_constant<int, 2>{}, q + 1, nw_slab, next_live);
                {
                    const __amdgpu_buffer_rsrc_t rw = make_rsrc_sgpr(w_ptr, w_bytes);
                    const unsigned b_adv = ((unsigned)t * (unsigned)CS + pf_slab) * 128u;
                    const unsigned base = lds0 + RING_OFF + (pf_step % 3u) * BSTAGE;
                    unsigned voff[BPW];
#pragma unroll
                    for (int j = 0; j < BPW; ++j) voff[j] = pvb[j] + b_adv;
#pragma unroll
                    for (int j = 0; j < BPW; ++j) asm volatile("" : "+v"(voff[j]));
#pragma unroll
                    for (int j = 0; j < BPW; ++j) piece(rw, base + (lw * BPW + j) * 1024, voff[j]);
                    ++pf_step;
                }
                ++k;
                W3_STAMP(0);
                if (k == 1) continue;      // (after the first issue step there is nothing to hand over yet)
                if (t >= 2 && t <= 7) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BPW + 2) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BPW) : "memory");
                W3_STAMP(1);
                __syncthreads();
                W3_STAMP(2);
                if (k >= 3 && ++ckt == 9u * (unsigned)CS) {
                    ckt = 0;
                    ++ci;
                    __syncthreads();                            // pause while the MFMA waves store the tile
                    W3_STAMP(3);
                }
            }
            // the issue cursor moves to the next slab; the window after the next one is then (tile, slab) + 1
            if (++pf_slab == (unsigned)CS) {
                pf_slab = 0;
                setup_weights(++pf_tile);
            }
            if (++nw_slab == (unsigned)CS) {
                nw_slab = 0;
                setup_window(++nw_tile);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W3_STAMP_FLUSH;
        return;
    }

    // =================================== MFMA waves 0..7 ===================================
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l16 = lane & 15, lq = lane >> 4;
    // per row block and kw: pre-swizzled byte offset of the lane's pixel row, k-group 0 (k-group 1 = the same ^ 64).  Recomputed
    // after every epilogue (from an opaque copy of the lane id, so the old values are DEAD across it): kept live they were what the
    // register allocator spilled, and the K loop re-read them from scratch every step.
    // per row block: pre-swizzled byte offsets of the lane's pixel row for kw = 0..2, k-group 0 (k-group 1 = the same ^ 64).  At
    // RBW = 7 the three are PACKED into one register (bits 0-15 the byte offset of window row w0, bits 16-24 the three chunk
    // positions lq ^ key(w0 + kw)): 21 offset registers there pushed the kernel into scratch, and the K loop re-read them every step.
    constexpr bool PACKED = RBW > 4;
    int a_off[RBW][PACKED ? 1 : 3];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
        int pi = (wm * RBW + rb) * 16 + l16;
        pi = pi < NPIX ? pi : NPIX - 1;                 // rows past the tile read a valid pixel and are never stored
        const int img = pi / (p.TR * p.W), rem = pi - img * (p.TR * p.W);
        const int yy = rem / p.W, xx = rem - yy * p.W;
        const int w0 = img * WROWS_IMG + yy * p.PITCH + xx;      // window row of tap (0, 0)
        if constexpr (PACKED) {
            a_off[rb][0] = (w0 * ROWB) | ((lq ^ swz_key(w0)) << 16) | ((lq ^ swz_key(w0 + 1)) << 19) | ((lq ^ swz_key(w0 + 2)) << 22);
        } else {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) a_off[rb][kw] = (w0 + kw) * ROWB + 16 * (lq ^ swz_key(w0 + kw));
        }
    }
    int b_g0[2], b_g1[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int brow = wn * 32 + 8 * (l16 >> 2) + 4 * nb + (l16 & 3);
        b_g0[nb] = brow * ROWB + 16 * (lq ^ swz_key(brow));
        b_g1[nb] = brow * ROWB + 16 * ((4 + lq) ^ swz_key(brow));
    }
    f32x4 acc[RBW][2];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    const int row0 = wm * RBW * 16 + l16;               // tile pixel of row block 0
    const unsigned yvoff = ((unsigned)row0 * (unsigned)p.Cout + (unsigned)(wn * 32 + 8 * lq)) * 2u;

    int tn0, ty0, c0;
    unsigned m0;
    unsigned ci = 0, g = 0;
    tile_origin(0, tn0, ty0, m0, c0);
    __syncthreads();                                        // (the loaders' first hand-over: window 0 and step 0 have landed)
    W3_STAMP_DECL;

    auto mfma_block = [&](int rb, const bf16x8& x0, const bf16x8& x1, const bf16x8& w00, const bf16x8& w01, const bf16x8& w10, const bf16x8& w11) __attribute__((always_inline)) {
        acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w00, x0, acc[rb][0], 0, 0, 0);
        acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w10, x0, acc[rb][1], 0, 0, 0);
        acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w01, x1, acc[rb][0], 0, 0, 0);
        acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w11, x1, acc[rb][1], 0, 0, 0);
    };

    unsigned slab_in_tile = 0;
    for (unsigned q = 0; q < nq; ++q) {
        const unsigned char* win = smem + (q & 1u) * WIN_BYTES;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t - 3 * (t / 3);
            const unsigned char* stg = smem + RING_OFF + (g % 3u) * BSTAGE;
            const unsigned char* wrow = win + kh * p.PITCH * ROWB;
            const bf16x8 b00 = *(const bf16x8*)(stg + b_g0[0]), b01 = *(const bf16x8*)(stg + b_g1[0]);
            const bf16x8 b10 = *(const bf16x8*)(stg + b_g0[1]), b11 = *(const bf16x8*)(stg + b_g1[1]);
            bf16x8 a0[RBW], a1[RBW];
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                int pk = a_off[rb][PACKED ? 0 : kw];
                asm volatile("" : "+v"(pk));    // opaque: hipcc otherwise precomputes all 9 x 2 RBW fragment addresses outside the loop (spills)
                const int o = PACKED ? (pk & 0xFFFF) + kw * ROWB + (((pk >> (16 + 3 * kw)) & 7) << 4) : pk;
                a0[rb] = *(const bf16x8*)(wrow + o);
                a1[rb] = *(const bf16x8*)(wrow + (o ^ 64));
            }
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) mfma_block(rb, a0[rb], a1[rb], b00, b01, b10, b11);
            // schedule: weight fragments + PRE row blocks of activation fragments up front, then per row block its four MFMAs with
            // the two reads of a later block in their shadow (without this hipcc hoists all 2 RBW reads: 56 live registers at
            // RBW = 7 and the pre-swizzled offsets were spilled to scratch and re-read INSIDE the loop)
            constexpr int PRE = 2;
            __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * PRE, 0);
#pragma unroll
            for (int grp = 0; grp < RBW; ++grp) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (grp + PRE < RBW) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            ++g;
            W3_STAMP(0);
            __syncthreads();                            // the next step's weights (and, at t == 8, the next window) have landed
            W3_STAMP(1);
        }
        if (++slab_in_tile == (unsigned)CS) {
            slab_in_tile = 0;
            const long long yorg = ((long long)m0 * p.Cout + c0) * 2ll, ybytes = ((long long)(p.M - m0) * p.Cout - c0) * 2ll;
            const __amdgpu_buffer_rsrc_t ry = make_rsrc_sgpr((char*)p.y + yorg, ybytes);
            const __amdgpu_buffer_rsrc_t rr = make_rsrc_sgpr((const char*)p.res + yorg, p.res ? ybytes : 0);
            f32x4 e_sc[2], e_sh[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                e_sc[nb] = *(const f32x4*)(smem + E_OFF + (ci & 1u) * 2048 + (wn * 32 + 8 * lq + 4 * nb) * 4);
                e_sh[nb] = *(const f32x4*)(smem + E_OFF + (ci & 1u) * 2048 + 1024 + 512 + (wn * 32 + 8 * lq + 4 * nb) * 4);
            }
            // residual rows two row blocks ahead of their use (not all RBW at once: registers)
            auto res_load = [&](int rb) __attribute__((always_inline)) {
                return bload16(rr, (row0 + 16 * rb) < NPIX ? yvoff : 0x80000000u, __builtin_amdgcn_readfirstlane((unsigned)(16 * rb) * (unsigned)p.Cout * 2u));
            };
            f32x4 rres[RBW];
            if (p.res) {
#pragma unroll
                for (int rb = 0; rb < 2 && rb < RBW; ++rb) rres[rb] = res_load(rb);
            }
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                if (p.res && rb + 2 < RBW) rres[rb + 2] = res_load(rb + 2);
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
                // rows past the tile's pixels belong to the NEXT tile: an out-of-range offset drops them
                bstore16_welded(o, ry, (row0 + 16 * rb) < NPIX ? yvoff : 0x80000000u,
                                __builtin_amdgcn_readfirstlane((unsigned)(16 * rb) * (unsigned)p.Cout * 2u));
            }
            zero_acc();
            tile_origin(++ci, tn0, ty0, m0, c0);
            W3_STAMP(2);
            __syncthreads();                            // lets the loaders go on
            W3_STAMP(3);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W3_STAMP_FLUSH;
}

struct Win3Cfg { int rbw, waves_m, tr, img, pitch; };

// Tile geometry for an H x W map and Cout channels; rbw == 0: not covered
Win3Cfg choose_win3(int n, int h, int w, int cout) {
    Win3Cfg best{0, 0, 0, 0, 0};
    const int waves_m = cout % 128 == 0 ? 2 : 4;
    const int rb_max = waves_m == 2 ? 7 : 4;
    const int pitch = (w + 2 + 15) / 16 * 16;
    if (pitch > 64) return best;
    double best_eff = 0;
    for (int tr = 1; tr <= h; ++tr) {
        if (h % tr) continue;
        for (int img = 1; img <= (tr == h ? 4 : 1); ++img) {
            const int npix = img * tr * w, wrows = img * (tr + 2) * pitch;
            if (wrows > 384 || npix > waves_m * rb_max * 16) continue;
            const int nb16 = (npix + 15) / 16;
            int rbw = (nb16 + waves_m - 1) / waves_m;
            rbw = waves_m == 2 ? (rbw <= 4 ? 4 : 7) : (rbw <= 2 ? 2 : 4);      // instantiated heights
            // useful share of the MFMA rows, with a small bonus for bigger tiles (fewer window halos, fewer epilogues)
            const double eff = (double)npix / (waves_m * rbw * 16) * (1.0 - 0.25 * (double)(wrows - npix) / wrows);
            if (eff > best_eff) { best_eff = eff; best = Win3Cfg{rbw, waves_m, tr, img, pitch}; }
        }
    }
    (void)n;
    return best;
}

HSEFR_KNOB(g_w3_off, 0);    // dev builds: 1 = never use this kernel, 2 = for every shape it covers

template <int RBW, int WAVES_M>
int launch_w3(Win3Params& p, hipStream_t s) {
    constexpr int BN = (8 / WAVES_M) * 32;
    p.tiles_n = (unsigned)(p.Cout / BN);
    const long long tiles_m = (long long)((p.N + p.IMG - 1) / p.IMG) * p.tiles_y;
    const long long total = tiles_m * p.tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv3x3_win: too many tiles");
    p.total_tiles = (unsigned)total;
    const unsigned g = (unsigned)(total < 256 ? total : 256);
    HSEFR_LAUNCH((conv3x3_win_bf16_kernel<RBW, WAVES_M>), dim3(g), dim3(768), 0, s, p);
    return launch_status("conv3x3_win_bf16");
}

}  // namespace

#ifdef HSEFR_DEV
int read_w3_stamps(void* host_out, size_t bytes) {
#ifdef HSEFR_CD_STAMPS
    HSEFR_REQUIRE(bytes <= sizeof(unsigned long long) * 256 * 12 * 8, HSEFR_ERR_INVALID, "read_w3_stamps: too many bytes");
    HSEFR_HIP_CHECK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w3_stamps), bytes));
    return HSEFR_OK;
#else
    (void)host_out; (void)bytes;
    set_error("read_w3_stamps: library built without -DHSEFR_CD_STAMPS");
    return HSEFR_ERR_UNSUPPORTED;
#endif
}
void set_w3_off(int v) { g_w3_off = v; }
#endif

bool conv3x3_win_forced() { return g_w3_off == 2; }

bool conv3x3_win_bf16_supported(long long n, int h, int w, int c, int cout) {
    if (g_w3_off == 1) return false;
    if (!(c > 0 && c % 64 == 0 && cout > 0 && cout % 64 == 0 && n > 0 && n * h * w * (long long)c * 2 < (1ll << 31) &&
          n * h * w * (long long)cout * 2 < (1ll << 32) && n * h * w < (1ll << 31) && 9ll * c * 2 * 128 < (1ll << 31)))
        return false;
    return choose_win3((int)n, h, w, cout).rbw > 0;
}

int launch_conv3x3_win_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y, int n, int h,
                            int w, int c, int cout, int act, hipStream_t s) {
    HSEFR_REQUIRE(conv3x3_win_bf16_supported(n, h, w, c, cout), HSEFR_ERR_UNSUPPORTED, "conv3x3_win_bf16: shape not covered");
    HSEFR_REQUIRE(act == HSEFR_ACT_NONE || act == HSEFR_ACT_RELU || act == HSEFR_ACT_RELU6, HSEFR_ERR_UNSUPPORTED, "conv3x3_win_bf16: act %d", act);
    const Win3Cfg cfg = choose_win3(n, h, w, cout);
    Win3Params p;
    p.x = x; p.wt = wt; p.scale = scale; p.shift = shift; p.res = res; p.y = y;
    p.x_bytes = (long long)n * h * w * c * 2;
    p.N = n; p.H = h; p.W = w; p.C = c; p.Cout = cout;
    p.TR = cfg.tr; p.IMG = cfg.img; p.PITCH = cfg.pitch; p.tiles_y = h / cfg.tr;
    p.act_lo = act == HSEFR_ACT_NONE ? -INFINITY : 0.f;
    p.act_hi = act == HSEFR_ACT_RELU6 ? 6.f : INFINITY;
    p.M = (unsigned)((long long)n * h * w);
    p.reverse = sweep_reverse();
    if (cfg.waves_m == 2) return cfg.rbw == 7 ? launch_w3<7, 2>(p, s) : launch_w3<4, 2>(p, s);
    return cfg.rbw == 4 ? launch_w3<4, 4>(p, s) : launch_w3<2, 4>(p, s);
}

}  // namespace hsefr
