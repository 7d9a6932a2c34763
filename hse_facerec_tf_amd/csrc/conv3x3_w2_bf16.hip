// 3x3 / stride 1 / pad 1 bf16 convolution + scale + shift (+ residual) + act, second window form (round 5): FOUR wide MFMA waves.
// The 3x3 layers of ResNet-50's bottlenecks (resnet50_ft, the graph behind vgg2_resnet.pb at facerec_test.py:213).  NHWC bf16 in / out,
// fp32 accumulation, gfx950.  Same results contract as conv_bf16.hip / conv_dma_bf16.hip / conv3x3_win_bf16.hip (rounding points of
// oracle/resnet50.py); the K order inside a 64-channel slab differs from theirs (other bits, same bar).
//
// Why another one (in-kernel stamps of round 5, profiles/r05_conv_stamps.txt): the im2col-by-DMA kernel is bound by the rate its
// four loader waves can issue 1-KiB LDS-DMA pieces (44 per K-step: 1780 of a step's 2040 cycles); the first window kernel moved
// a third of that but its eight MFMA waves (wave tile 112 x 32) then spent 1340 cycles per step where the matrix pipe needs 896:
// 144 KB of fragment reads per step (576 LDS cycles) + 2-way bank conflicts on every tap that is not a multiple of four pixels
// (the row-keyed swizzle) + ~50 address VALU per wave and step (packed pre-swizzled offsets, unpacked per fragment).  Here:
//   * 4 MFMA waves (one per SIMD) with wave tiles of (16 RB) x 64: 224 x 128 per workgroup as 2 x 2 -- 90 KB of fragment reads per
//     step instead of 144 -- + 4 loader waves; 8 waves per CU = 256 registers each: 112 accumulators fit;
//   * PADDED-ROW pixel mapping: an MFMA row block is 16 consecutive columns of ONE image row (14-pixel rows: 2 columns computed
//     and dropped; 28: two blocks per row; 56: four), so a fragment read is 16 consecutive window pixels for every tap and every
//     row block, and the tap / row-block displacement is an INSTRUCTION IMMEDIATE: three address registers per wave (one per kw),
//     no address arithmetic in the K loop;
//   * a swizzle that is conflict-free for ANY 16 consecutive pixels: chunk ^ (x & 6) (found by exhaustive search against the
//     ds_read_b128 lane groups of the MI355X guide; the usual ((row >> 1) & 7) ^ ((row & 1) << 2) is conflict-free only for starts
//     that are multiples of 4);
//   * a FOUR-stage weight ring published ONE STEP EARLY (the loaders run three steps ahead): during step g the MFMA wave already
//     reads step g + 1's first fragments -- a single wave per SIMD has no partner to hide the LDS latency behind the step barrier;
//     weight fragments are re-loaded into the registers of the ones they replace right after their last use;
//   * FLAT geometry for maps of at most 7 x 7 pixels (the last stage): a tile is IMG whole images as a flat list of 8-pixel rows --
//     [zero row][7 image rows][zero row][7 image rows] ... with a zero cell in front of every row -- so that the zero row under one
//     image is the zero row over the next, the zero cell in front of a row is the right neighbour of the row before it, and output
//     slot s = 64 img + 8 y + x reads window cell s + 8 kh + kw for EVERY tap: the same affine addressing (49 of 64 slots are pixels);
//   * everything else as before: LDS-DMA with out-of-range offsets for the padding, weights first (a lane owns 8 consecutive
//     channels: 16-byte stores straight from the accumulators), one barrier per step, persistent workgroups.
// Every output element is accumulated in one fixed order by one wave: bit-identical run to run, independent of the grid.
#include <type_traits>

#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ROWB = 128;                 // bytes per LDS row: 64 bf16 = one K-step of one pixel / one weight row
constexpr int NSTG = 4;                   // weight ring stages

#ifdef HSEFR_CD_STAMPS
__device__ unsigned long long g_w2_stamps[256 * 8 * 8];
#define W2_STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); st[i] += _t - tprev; tprev = _t; } while (0)
#define W2_STAMP_DECL unsigned long long st[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tprev = __builtin_amdgcn_s_memtime(); const unsigned long long tstart = tprev
#define W2_STAMP_FLUSH do { if (lane == 0 && blockIdx.x < 256) { unsigned long long* o = g_w2_stamps + (blockIdx.x * 8 + wave) * 8; \
    for (int i_ = 0; i_ < 6; ++i_) o[i_] = st[i_]; o[6] = __builtin_amdgcn_s_memtime() - tstart; o[7] = nsteps; } } while (0)
#else
#define W2_STAMP(i) do { } while (0)
#define W2_STAMP_DECL do { } while (0)
#define W2_STAMP_FLUSH do { } while (0)
#endif

__device__ __forceinline__ unsigned f2bf_bits(float f) { return hsefr_bf16_bits(f); }      // round-to-nearest-even (common.h)
__device__ __forceinline__ float bfround(float f) { return __uint_as_float(f2bf_bits(f) << 16); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_sgpr(const void* ptr, long long bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned n = __builtin_amdgcn_readfirstlane(bytes <= 0 ? 0u : (bytes > 0xffffffffll ? 0xffffffffu : (unsigned)bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}
// the workgroup barrier WITHOUT __syncthreads()'s waits: LDS reads of the next step may stay in flight across it (see the header);
// the "memory" clobber keeps hipcc from moving LDS accesses over it
__device__ __forceinline__ void step_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct W2Params {
    const void* x;       // [N,H,W,C] bf16
    const void* wt;      // [Cout][9*C] bf16, k = (kh*3 + kw)*C + c
    const float* scale;
    const float* shift;
    const void* res;     // [M,Cout] bf16 or null
    void* y;             // [M,Cout] bf16
    long long x_bytes;
    int N, H, W, C, Cout;
    int tiles_y;         // ceil(H / TR)
    float act_lo, act_hi;
    unsigned M;
    unsigned tiles_n, total_tiles;
    int reverse;
};

// RB: 16-row blocks per MFMA wave; WAVES_M x (4 / WAVES_M) MFMA waves; the WAVES_M waves are WX across x and WAVES_M / WX down y;
// a wave's RB blocks are RBX across x and RB / RBX down y.  FLAT: the geometry for maps of at most 7 x 7 (WX = RBX = 1 unused).
template <int RB, int WAVES_M, int WX, int RBX, bool FLAT = false>
__global__ __launch_bounds__(512) void conv3x3_w2_bf16_kernel(W2Params p) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int BN = WAVES_N * 64;
    constexpr int BPW = BN / 32;                    // weight pieces per loader wave and step
    constexpr int BSTAGE = BN * ROWB;
    constexpr int WY = WAVES_M / WX, RBY = RB / RBX;
    static_assert(WY * WX == WAVES_M && RBY * RBX == RB, "wave / row-block grids");
    constexpr int TR = WY * RBY;                    // image rows per tile
    constexpr int CBX = WX * RBX;                   // 16-column blocks per image row
    constexpr int PX = FLAT ? 8 : 16 * CBX + 2;     // window pitch in pixels: columns -1 .. 16 CBX (FLAT: the zero cell + 7 pixels)
    constexpr int IMG = WAVES_M * RB / 4;           // FLAT: images per tile (64 slots = 4 row blocks each)
    static_assert(!FLAT || (WAVES_M * RB) % 4 == 0, "FLAT tiles hold whole images");
    constexpr int NWR = FLAT ? IMG * 64 + 24 : (TR + 2) * PX;      // window pixels (FLAT: + the cells the last slots' taps reach)
    constexpr int WPIECES = (NWR + 7) / 8;
    constexpr int WIN_BYTES = WPIECES * 1024;
    constexpr int WSLOTS = (WPIECES + 3) / 4;       // window pieces per loader wave
    constexpr int WPT = (WSLOTS + 6) / 7;           // ... issued per tap-step, taps 0..6
    constexpr int RING_OFF = 2 * WIN_BYTES;
    constexpr int E_OFF = RING_OFF + NSTG * BSTAGE;
    constexpr int DUMMY_OFF = E_OFF + 4096;         // 1 KiB that absorbs the pieces issued only to keep the counts fixed
    static_assert(DUMMY_OFF + 1024 <= 160 * 1024, "LDS budget");
    static_assert((FLAT ? 16 * (RB - 1) + 16 : (TR + 1) * PX + 16 * (RBX - 1)) * ROWB + 64 < 65536, "tap / row-block displacements are 16-bit immediates");
    static_assert((9 * RB) % 3 == 0, "the fragment ring keeps its phase from slab to slab");
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

    // tile i of this workgroup -> image, first row, first output channel
    auto tile_origin = [&](unsigned i, int& tn, int& ty0, int& cc0) __attribute__((always_inline)) {
        const unsigned lt = xcd_remap_dir(blockIdx.x + (i < ntile ? i : ntile - 1) * gridDim.x, p.total_tiles, p.reverse);
        const unsigned tm = lt / p.tiles_n;
        cc0 = (int)(lt - tm * p.tiles_n) * BN;
        if (FLAT) { tn = (int)tm * IMG; ty0 = 0; return; }
        tn = (int)(tm / (unsigned)p.tiles_y);
        ty0 = (int)(tm - (unsigned)tn * (unsigned)p.tiles_y) * TR;
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
        // weight pieces: LDS rows 8 (lw BPW + j) .. + 7 of the stage.  LDS row R = 64 wn + 16 nb + i holds output channel
        // 64 wn + 32 (nb >> 1) + 8 (i >> 2) + 4 (nb & 1) + (i & 3): with the weights as the first MFMA operand a lane then owns
        // 8 consecutive channels per pair of channel blocks (16-byte stores from the accumulators)
        unsigned pvb[BPW];
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int R = (lw * BPW + j) * 8 + (lane >> 3);
            const int nb = (R >> 4) & 3, i = R & 15;
            const int ch = (R & ~63) + 32 * (nb >> 1) + 8 * (i >> 2) + 4 * (nb & 1) + (i & 3);
            pvb[j] = (unsigned)ch * wrowbytes + 16u * (unsigned)((lane & 7) ^ (R & 6));
        }
        // window pieces of this wave: piece lw + 4 s, s = 0 .. WSLOTS - 1 (those >= WPIECES do not exist)
        unsigned wbase[WSLOTS];
        auto setup_window = [&](unsigned i) __attribute__((always_inline)) {     // slab-0 offsets of tile i's window
            int tn, ty0, cc0;
            tile_origin(i, tn, ty0, cc0);
#pragma unroll
            for (int s = 0; s < WSLOTS; ++s) {
                const int pw = lw + 4 * s;
                const int w = pw * 8 + (lane >> 3);
                int wy, wx, img = 0;
                if (FLAT) { img = w >> 6; wy = (w >> 3) & 7; wx = w & 7; }      // row 0 of an image's eight = the shared zero row
                else { wy = w / PX; wx = w - wy * PX; }
                const int iy = ty0 - 1 + wy, ix = wx - 1;
                const bool ok = w < NWR && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && tn + img < p.N && (!FLAT || img < IMG);
                wbase[s] = ok ? (unsigned)(((tn + img) * p.H + iy) * p.W + ix) * (unsigned)(p.C * 2) + 16u * (unsigned)((lane & 7) ^ (wx & 6)) : 0x80000000u;
            }
        };
        // issue window slots [s0, s0 + cnt) of the window with sequence number wq (buffer wq & 1), slab offset wsl; slots that do not
        // exist go to the dummy KiB with an out-of-range source (zero fill): the number of pieces per step stays fixed
        auto issue_window = [&](int s0, auto CNT, unsigned wq, unsigned wsl, bool live) __attribute__((always_inline)) {
            constexpr int cnt = decltype(CNT)::value;
            unsigned voff[cnt];
#pragma unroll
            for (int s = 0; s < cnt; ++s) voff[s] = (live && s0 + s < WSLOTS) ? wbase[s0 + s < WSLOTS ? s0 + s : 0] + wsl * 128u : 0x80000000u;
#pragma unroll
            for (int s = 0; s < cnt; ++s) asm volatile("" : "+v"(voff[s]));
#pragma unroll
            for (int s = 0; s < cnt; ++s) {
                const int pw = lw + 4 * (s0 + s);
                const unsigned dst = (live && s0 + s < WSLOTS && pw < WPIECES) ? lds0 + (wq & 1u) * WIN_BYTES + (unsigned)pw * 1024u : lds0 + DUMMY_OFF;
                piece(rx, dst, voff[s]);
            }
        };
        // weight cursor: the step whose weights go out next (three ahead of the step the MFMA waves are on)
        const char* w_ptr = nullptr;
        long long w_bytes = 0;
        unsigned pf_tile = 0, pf_slab = 0, pf_tap = 0, pf_step = 0;
        auto setup_weights = [&](unsigned i) __attribute__((always_inline)) {
            int tn, ty0, cc0;
            tile_origin(i, tn, ty0, cc0);
            w_ptr = (const char*)p.wt + (long long)cc0 * wrowbytes;
            w_bytes = (long long)(p.Cout - cc0) * wrowbytes;
        };
        auto issue_weights = [&]() __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t rw = make_rsrc_sgpr(w_ptr, w_bytes);
            const unsigned b_adv = (pf_tap * (unsigned)CS + pf_slab) * 128u;
            const unsigned base = lds0 + RING_OFF + (pf_step & (NSTG - 1)) * BSTAGE;
            const bool live = pf_step < nsteps;
            unsigned voff[BPW];
#pragma unroll
            for (int j = 0; j < BPW; ++j) voff[j] = live ? pvb[j] + b_adv : 0x80000000u;
#pragma unroll
            for (int j = 0; j < BPW; ++j) asm volatile("" : "+v"(voff[j]));
#pragma unroll
            for (int j = 0; j < BPW; ++j) piece(rw, base + (lw * BPW + j) * 1024, voff[j]);
            ++pf_step;
            if (++pf_tap == 9u) {
                pf_tap = 0;
                if (++pf_slab == (unsigned)CS) {
                    pf_slab = 0;
                    setup_weights(++pf_tile);
                }
            }
        };

        // ---- prologue: window 0 completely and the weights of steps 0, 1, 2 ----
        setup_window(0);
        issue_window(0, std::integral_constant<int, WSLOTS>{}, 0u, 0u, true);
        unsigned nw_tile = CS > 1 ? 0u : 1u, nw_slab = CS > 1 ? 1u : 0u;      // (tile, slab) of the NEXT window (sequence number q + 1)
        if (CS == 1) setup_window(1);
        setup_weights(0);
        issue_weights();
        issue_weights();
        issue_weights();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        step_barrier();                                        // P: window 0 and steps 0..2 have landed
        unsigned ci = 0;                                        // tile the MFMA waves are on (for the epilogue constants)
        unsigned slab_in_tile = 0;
        W2_STAMP_DECL;
        for (unsigned q = 0; q < nq; ++q) {
            const bool next_live = q + 1 < nq;                  // window q + 1 exists
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t == 0 && slab_in_tile == 0 && lw == 3) {
                    // the tile's epilogue constants by LDS-DMA: scale[c0 ..] as lanes 0-31 of one piece, shift[..] as lanes 32-63 of a
                    // second one; issued AHEAD of this step's pieces so the counted wait of the NEXT step covers them
                    int tn, ty0, e_c0;
                    tile_origin(ci, tn, ty0, e_c0);
                    const __amdgpu_buffer_rsrc_t rd = make_rsrc_sgpr(p.scale + e_c0, (long long)(p.Cout - e_c0) * 4),
                                                 rs = make_rsrc_sgpr(p.shift + e_c0, (long long)(p.Cout - e_c0) * 4);
                    const unsigned eb = lds0 + E_OFF + (ci & 1u) * 2048u;
                    piece(rd, eb, lane < 32 ? 16u * lane : 0x80000000u);
                    piece(rs, eb + 1024, lane >= 32 ? 16u * (unsigned)(lane - 32) : 0x80000000u);
                }
                // window q + 1 during taps 0..6: its buffer was last read in the final step of slab q - 1, behind the barrier this
                // iteration started from; it is complete at the barrier that ends tap 7 (the wait of tap 7 covers tap 6's pieces),
                // one step before its first fragment is read
                if (t <= 6) issue_window(WPT * t, std::integral_constant<int, WPT>{}, q + 1, nw_slab, next_live);
                issue_weights();                                // step 9 q + t + 3
                W2_STAMP(0);
                if (t <= 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BPW + WPT) : "memory");     // everything older than this iteration's pieces
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BPW) : "memory");
                W2_STAMP(1);
                step_barrier();                                 // B_g: step g + 2 has landed, the slots of step g are free
                W2_STAMP(2);
            }
            if (++slab_in_tile == (unsigned)CS) { slab_in_tile = 0; ++ci; }
            if (++nw_slab == (unsigned)CS) {
                nw_slab = 0;
                setup_window(++nw_tile);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W2_STAMP_FLUSH;
        return;
    }

    // =================================== MFMA waves 0..3 ===================================
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l16 = lane & 15, lq = lane >> 4;
    const int y0w = FLAT ? 0 : (wm / WX) * RBY, x0w = FLAT ? 16 * RB * wm : 16 * RBX * (wm % WX);      // the wave's first row block: tile row, column (FLAT: slot)
    // fragment addresses: window pixel (y0w + dy + kh, x0w + dx + l16 + kw) [window coordinates: image column + 1], 16-byte chunk
    // (4 half + lq) ^ ((l16 + kw) & 6): one register per kw (half 1 = the same ^ 64), (dy + kh, dx) is an immediate
    unsigned acur[2][3], anext[2][3];                                   // [half][kw]
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        acur[0][kw] = (unsigned)((y0w * PX + x0w + l16 + kw) * ROWB + 16 * (lq ^ ((l16 + kw) & 6)));
        acur[1][kw] = acur[0][kw] ^ 64u;
        anext[0][kw] = acur[0][kw] + WIN_BYTES;
        anext[1][kw] = acur[1][kw] + WIN_BYTES;
    }
    const unsigned b0 = (unsigned)(RING_OFF + (wn * 64 + l16) * ROWB + 16 * (lq ^ (l16 & 6)));

    f32x4 acc[RB][4];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    int tn, ty0, c0;
    unsigned ci = 0, g = 0;
    tile_origin(0, tn, ty0, c0);

    bf16x8 bfr[4][2];        // weight fragments of the current step (re-loaded one by one for the next step in its last block)
    bf16x8 ar[3][2];         // activation fragments: ring over row blocks, two blocks ahead of the MFMAs
    // (the half-1 address is a register of its own: (a + imm) ^ 64 would keep the displacement out of the instruction's offset field)
    auto lda = [&](const unsigned (&ab)[2][3], int kw, int imm, int half) __attribute__((always_inline)) -> bf16x8 {
        return *(const bf16x8*)(smem + ab[half][kw] + imm);
    };
    auto ldb = [&](const unsigned (&bb)[2], int nb, int half) __attribute__((always_inline)) -> bf16x8 {
        return *(const bf16x8*)(smem + bb[half] + nb * 16 * ROWB);
    };
    // displacement of row block rb for tap row kh, bytes
    auto aimm = [](int rb, int kh) constexpr { return FLAT ? (16 * rb + 8 * kh) * ROWB : ((rb / RBX + kh) * PX + 16 * (rb % RBX)) * ROWB; };

    step_barrier();                                         // P
    {
        const unsigned bfirst[2] = {b0, b0 ^ 64u};
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { bfr[nb][0] = ldb(bfirst, nb, 0); bfr[nb][1] = ldb(bfirst, nb, 1); }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) { ar[blk][0] = lda(acur, 0, aimm(blk, 0), 0); ar[blk][1] = lda(acur, 0, aimm(blk, 0), 1); }
    }
    W2_STAMP_DECL;

    unsigned slab_in_tile = 0;
    for (unsigned q = 0; q < nq; ++q) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const unsigned bn0 = b0 + ((g + 1u) & (NSTG - 1)) * BSTAGE;
            const unsigned bnext[2] = {bn0, bn0 ^ 64u};
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int blk = t * RB + rb;                // block number inside the slab
                // the block two ahead: same step, next step, or the next slab's first step (other window buffer)
                {
                    const int b2 = blk + 2, t2 = b2 / RB, rb2 = b2 - t2 * RB;
                    if (t2 < 9) {
                        ar[b2 % 3][0] = lda(acur, t2 % 3, aimm(rb2, t2 / 3), 0);
                        ar[b2 % 3][1] = lda(acur, t2 % 3, aimm(rb2, t2 / 3), 1);
                    } else {
                        ar[b2 % 3][0] = lda(anext, 0, aimm(rb2, 0), 0);
                        ar[b2 % 3][1] = lda(anext, 0, aimm(rb2, 0), 1);
                    }
                }
                const bf16x8 x0 = ar[blk % 3][0], x1 = ar[blk % 3][1];
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
                // schedule: the two activation reads under the block's first MFMAs; in a step's last block one weight read behind
                // each MFMA
                if (rb == RB - 1) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                }
            }
            ++g;
            W2_STAMP(0);
            step_barrier();                                 // B_g
            W2_STAMP(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next slab reads the other window buffer
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int h = 0; h < 2; ++h) { const unsigned tmp = acur[h][kw]; acur[h][kw] = anext[h][kw]; anext[h][kw] = tmp; }
        if (++slab_in_tile == (unsigned)CS) {
            slab_in_tile = 0;
            // ---- epilogue: lane (l16, lq) holds, per row block, pixel l16 x channels 64 wn + 32 j + 8 lq .. + 7 (j = 0, 1) ----
            const long long yorg = FLAT ? (long long)tn * p.H * p.W * p.Cout * 2ll + (long long)(c0 + wn * 64) * 2ll
                                        : (((long long)tn * p.H + ty0 + y0w) * p.W + x0w) * p.Cout * 2ll + (long long)(c0 + wn * 64) * 2ll;
            const long long ybytes = (long long)p.M * p.Cout * 2ll - yorg;
            const __amdgpu_buffer_rsrc_t ry = make_rsrc_sgpr((char*)p.y + yorg, ybytes);
            const __amdgpu_buffer_rsrc_t rr = make_rsrc_sgpr((const char*)p.res + yorg, p.res ? ybytes : 0);
            f32x4 e_sc[4], e_sh[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int cch = wn * 64 + 32 * (v >> 1) + 8 * lq + 4 * (v & 1);
                e_sc[v] = *(const f32x4*)(smem + E_OFF + (ci & 1u) * 2048 + cch * 4);
                e_sh[v] = *(const f32x4*)(smem + E_OFF + (ci & 1u) * 2048 + 1024 + 512 + cch * 4);
            }
            // FLAT: a block is two 8-slot rows of one image: lane -> (row l16 >> 3, column l16 & 7)
            const unsigned ylane = (FLAT ? (unsigned)((l16 >> 3) * p.W + (l16 & 7)) : (unsigned)l16) * (unsigned)p.Cout * 2u + 16u * (unsigned)lq;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int dy = rb / RBX, dx = 16 * (rb % RBX);
                const int fb = wm * RB + rb;                 // FLAT: block of the tile -> image fb / 4, rows 2 (fb % 4) and + 1
                const bool ok = FLAT ? ((l16 & 7) < p.W && 2 * (fb & 3) + (l16 >> 3) < p.H && tn + (fb >> 2) < p.N)
                                     : ((x0w + dx + l16) < p.W && (ty0 + y0w + dy) < p.H);
                const unsigned soff = __builtin_amdgcn_readfirstlane(FLAT ? (unsigned)((fb >> 2) * p.H * p.W + 2 * (fb & 3) * p.W) * (unsigned)p.Cout * 2u
                                                                          : (unsigned)(dy * p.W + dx) * (unsigned)p.Cout * 2u);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned voff = ok ? ylane + 64u * (unsigned)j : 0x80000000u;
                    float v[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * h + e] = fmaf(acc[rb][2 * j + h][e], e_sc[2 * j + h][e], e_sh[2 * j + h][e]);
                    if (p.res) {
                        const f32x4 rres = bload16(rr, voff, soff);
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const unsigned rw2 = __float_as_uint(rres[d]);
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
                    bstore16_welded(o, ry, voff, soff);
                }
            }
            zero_acc();
            tile_origin(++ci, tn, ty0, c0);
            W2_STAMP(2);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // stores issued from asm: drained before the wave ends
    W2_STAMP_FLUSH;
}

HSEFR_KNOB(g_w2_off, 0);    // dev builds: 1 = never use this kernel, 2 = for every shape it covers

template <int RB, int WAVES_M, int WX, int RBX, bool FLAT = false>
int launch_w2(W2Params& p, hipStream_t s) {
    constexpr int BN = (4 / WAVES_M) * 64;
    constexpr int TR = (WAVES_M / WX) * (RB / RBX);
    constexpr int IMG = WAVES_M * RB / 4;
    p.tiles_n = (unsigned)(p.Cout / BN);
    p.tiles_y = FLAT ? 1 : (p.H + TR - 1) / TR;
    const long long total = (FLAT ? (long long)((p.N + IMG - 1) / IMG) : (long long)p.N * p.tiles_y) * p.tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "conv3x3_w2: too many tiles");
    p.total_tiles = (unsigned)total;
    const unsigned g = (unsigned)(total < 256 ? total : 256);
    HSEFR_LAUNCH((conv3x3_w2_bf16_kernel<RB, WAVES_M, WX, RBX, FLAT>), dim3(g), dim3(512), 0, s, p);
    return launch_status("conv3x3_w2_bf16");
}

// 0: not covered; 1: rows of <= 16 pixels, 14 per tile (14 x 14 maps); 2: <= 32 pixels, 7 rows (28 x 28); 3: <= 64 pixels, 4 rows x 64 channels (56 x 56);
// 4: maps of at most 7 x 7, four images x 64 channels per tile (FLAT)
int w2_config(int h, int w, int cout) {
    if (w <= 7 && h <= 7) return 4;
    if (w <= 16 && cout % 128 == 0) return 1;
    if (w <= 32 && cout % 128 == 0) return 2;
    if (w <= 64 && cout % 64 == 0) return 3;
    return 0;
}

}  // namespace

#ifdef HSEFR_DEV
int read_w2_stamps(void* host_out, size_t bytes) {
#ifdef HSEFR_CD_STAMPS
    HSEFR_REQUIRE(bytes <= sizeof(unsigned long long) * 256 * 8 * 8, HSEFR_ERR_INVALID, "read_w2_stamps: too many bytes");
    HSEFR_HIP_CHECK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w2_stamps), bytes));
    return HSEFR_OK;
#else
    (void)host_out; (void)bytes;
    set_error("read_w2_stamps: library built without -DHSEFR_CD_STAMPS");
    return HSEFR_ERR_UNSUPPORTED;
#endif
}
void set_w2_off(int v) { g_w2_off = v; }
#endif

bool conv3x3_w2_forced() { return g_w2_off == 2; }

bool conv3x3_w2_bf16_supported(long long n, int h, int w, int c, int cout) {
    if (g_w2_off == 1) return false;
    if (!(c > 0 && c % 64 == 0 && cout > 0 && cout % 64 == 0 && n > 0 && h > 0 && w > 0 && n * h * w * (long long)c * 2 < (1ll << 31) &&
          n * h * w * (long long)cout * 2 < (1ll << 31) && n * h * w < (1ll << 31) && 9ll * c * 2 * 128 < (1ll << 31)))
        return false;
    return w2_config(h, w, cout) != 0;
}

// the shapes it is the measured-best kernel for (tools/kbench_conv.py, batch 128)
bool conv3x3_w2_bf16_preferred(int h, int w, int cout) {
    const int cfg = w2_config(h, w, cout);
    if (cfg == 1) return w >= 12 && h >= 12;
    if (cfg == 2) return w >= 24;
    if (cfg == 3) return w >= 48;
    if (cfg == 4) return w >= 6 && h >= 6;
    return false;
}

int launch_conv3x3_w2_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y, int n, int h,
                           int w, int c, int cout, int act, hipStream_t s) {
    HSEFR_REQUIRE(conv3x3_w2_bf16_supported(n, h, w, c, cout), HSEFR_ERR_UNSUPPORTED, "conv3x3_w2_bf16: shape not covered");
    HSEFR_REQUIRE(act == HSEFR_ACT_NONE || act == HSEFR_ACT_RELU || act == HSEFR_ACT_RELU6, HSEFR_ERR_UNSUPPORTED, "conv3x3_w2_bf16: act %d", act);
    W2Params p;
    p.x = x; p.wt = wt; p.scale = scale; p.shift = shift; p.res = res; p.y = y;
    p.x_bytes = (long long)n * h * w * c * 2;
    p.N = n; p.H = h; p.W = w; p.C = c; p.Cout = cout;
    p.act_lo = act == HSEFR_ACT_NONE ? -INFINITY : 0.f;
    p.act_hi = act == HSEFR_ACT_RELU6 ? 6.f : INFINITY;
    p.M = (unsigned)((long long)n * h * w);
    p.reverse = sweep_reverse();
    switch (w2_config(h, w, cout)) {
        case 1: return launch_w2<7, 2, 1, 1>(p, s);
        case 2: return launch_w2<7, 2, 2, 1>(p, s);
        case 4: return launch_w2<4, 4, 1, 1, true>(p, s);
        default: return launch_w2<4, 4, 1, 4>(p, s);
    }
}

}  // namespace hsefr
