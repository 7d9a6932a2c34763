// Pointwise (1x1) convolution + shift + act on PRE-SPLIT activations: the split-f16 GEMM of pwconv_f16s.hip with the
// VALU and the LDS writes taken out of its K loop.  NHWC, fp32 results, gfx950.
//
// Same graph nodes (Conv2D 1x1 -> Add shift -> Relu -> Minimum 6 -> Maximum 0, run by tf_sess.run at facerec_test.py:120 /
// facial_analysis.py:109) and the same arithmetic as pwconv_f16s.hip: every fp32 product a*w is formed as
// ah*wh + ah*wl + al*wh from two-term f16 splits of both operands, three f16 MFMAs into one fp32 accumulator.
//
// What changes is WHERE the activation is split.  pwconv_f16s.hip reads fp32 activations and splits them on their way
// into LDS: per K-step and thread 4 loads -> ~60 VALU -> 8 ds_write_b64, and round-1 counters put the LDS at ~85 % busy
// (reads + writes) under a matrix pipe that was 33 % busy.  Here the PRODUCER of the activation (the depthwise kernel,
// dwconv.hip with a_log2 > 0) stores it already split -- "split rows", the weight image's own format:
//
//     per pixel and 32-channel group one 128-byte row  [ hi(32 x f16) | lo(32 x f16) ],  hi = f16(x * 2^a_log2),
//     lo = f16(x * 2^a_log2 - hi)          (the same 4 bytes per element as fp32: byte-compatible tensor sizes)
//
// so BOTH operands of the GEMM go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave instruction,
// source chunks permuted per lane so that the linear LDS image is the swizzled one): no VGPR on the way in, no VALU and
// no ds_write in the loop.  The K loop is: DMA issue for the step after next, ds_read_b128 fragments, MFMAs, one barrier.
//
// Shape of the kernel
//   * ONE persistent workgroup of 12 waves per CU -- 8 MFMA waves and 4 loader waves that issue every DMA piece (a piece
//     costs its wave 100+ cycles of issue stall with the queues busy; in a first version where the MFMA waves issued their
//     own pieces right behind the step barrier both waves of a SIMD stalled together: 58 us per 36864 x 512 x 512 layer,
//     half the matrix rate) -- tile (32 * MB) x 128 with MB = 8 or 9 (256 or 288 rows): every layer
//     of MobileNet-192 at batch 256 has M = 9 * 2^k rows, so 288-row tiles fill whole rounds of 256 workgroups
//     (36864 x 512 -> 512 tiles = 2.0 rounds, where 128 x 128 tiles gave 2.25 rounds that cost 3); 224-pixel inputs
//     (M = 49 * 2^k) take whichever of the two needs fewer tile-rounds;
//   * three LDS stages of (BM + 128) x 128 B (156 KB for MB = 9): the DMA runs TWO K-steps ahead of the MFMAs;
//   * MFMA waves as 2 (M) x 4 (N), wave tile (16 * MB) x 32 on v_mfma_f32_16x16x32_f16 (one instruction per 32-deep K-step and
//     product; it sustains a higher clock than 32x32x16 on this chip -- DESIGN.md lesson 10), operands swapped so a lane
//     owns 4 consecutive output channels, 72 accumulator registers;
//   * epilogue straight from the accumulators: a lane's 4 consecutive channels are one 16-byte store, an instruction covers
//     16 rows x 64 B and the neighbouring channel block completes the lines (no LDS round trip, no barrier between tiles).
//
// DWM != 0: the NEXT block's depthwise 3x3 (stride 1, SAME) + scale + shift + ReLU6 runs in this kernel's epilogue and the tile
// leaves as that layer's SPLIT ROWS -- the pointwise result never exists in HBM and the depthwise kernel disappears (MobileNet's
// 12x12x512 and 6x6x1024 blocks: a 288-row tile is 2 or 8 WHOLE images, so the 3x3 neighbourhoods never leave the tile).  The
// epilogue walks the tile's four 32-channel chunks: the two MFMA waves that own chunk c park their activated results in the LDS
// stage the last K-step just released (288 rows x 128 B), then ALL TWELVE waves -- the loaders are idle here anyway -- compute the
// depthwise from LDS, lane = (pixel, channel quad), borders through a zero row, constants fetched once per tile by LDS-DMA into
// the same stage, and store 16-byte halves of split rows (the DPP pairing of dwconv.hip).  Two barriers per chunk.
//
// Every output element is accumulated over K in ONE fixed order by ONE wave: results are bit-identical run to run and
// independent of the grid.  (They are NOT bit-identical to pwconv_f16s.hip: a 16x16x32 MFMA sums its 32 products in a
// different internal order than two 32x32x16 steps -- same error bound, same 2e-6 test bar.)
#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef HSEFR_PS_STAMPS
// Diagnostic build only (HSEFR_DEV=1 HSEFR_EXTRA_FLAGS=-DHSEFR_PS_STAMPS=1 build.sh): per-wave s_memtime sums of the step phases.
__device__ unsigned long long g_ps_stamps[256 * 12 * 8];
#if HSEFR_PS_STAMPS == 2      // -DHSEFR_PS_STAMPS=2: only the waves' lifetimes (no stamp inside the loops: DESIGN.md lesson 56)
#define PS_STAMP(i) do { } while (0)
#else
#define PS_STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); st[i] += _t - tprev; tprev = _t; } while (0)
#endif
#define PS_STAMP_DECL unsigned long long st[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tprev = __builtin_amdgcn_s_memtime(); const unsigned long long tstart = tprev
#define PS_STAMP_FLUSH do { if (lane == 0 && blockIdx.x < 256) { unsigned long long* o = g_ps_stamps + (blockIdx.x * 12 + wave) * 8; \
    for (int i_ = 0; i_ < 6; ++i_) o[i_] = st[i_]; o[6] = __builtin_amdgcn_s_memtime() - tstart; o[7] = nsteps; } } while (0)
#else
#define PS_STAMP(i) do { } while (0)
#define PS_STAMP_DECL do { } while (0)
#define PS_STAMP_FLUSH do { } while (0)
#endif

#if defined(HSEFR_PS_KO) && ((HSEFR_PS_KO) & 4)
// (the value stays "used": with nothing behind it hipcc removes the depthwise arithmetic that produced it, and the build measures the
// whole epilogue -- bit 8 asks for exactly that)
#define PS_EPI_STORE(v, r, off) do { if (!((HSEFR_PS_KO) & 8)) { f32x4 v_ = (v); asm volatile("" ::"v"(v_)); } } while (0)
#else
#define PS_EPI_STORE(v, r, off) bstore16_welded_nt(v, r, off, 0u)
#endif
constexpr int ROWB = 128;       // bytes per split row: 32 hi halves | 32 lo halves
constexpr int BN = 128;

// swizzle key of a tile row: the 16-B chunk c of row r lives at chunk position c ^ key(r) (pwconv_f16s.hip's swizzle)
__device__ __forceinline__ int swz_key(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 2); }

// the depthwise layer fused behind the GEMM (DW kernels): dwc = [11][Cout] floats (taps 0..8 row-major, then scale and shift, the
// latter two already multiplied by out_scale), W = H = map edge, ys = its split-row output
struct PsDwParams {
    const float* dwc;
    void* ys;
    int W, HW;
    float clamp_hi;       // 6 * out_scale (pool mode: 1 / HW)
    int tile_rows;        // rows of the matrix a tile ADVANCES by = whole maps per tile x HW (<= 32 MB; the tile's remaining rows are
                          // computed but belong to the next tile: 14 x 14 maps ride in 224-row tiles, five 7 x 7 maps in 256-row ones)
};

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MB, int ACT, int DWM, int MW = 12>
__global__ __launch_bounds__(768, 1) void pwconv_ps_kernel(const void* __restrict__ xs, const void* __restrict__ wsplit,
                                                           const float* __restrict__ descale, const float* __restrict__ shift,
                                                           float* __restrict__ y, long long M, int K, int Cout, unsigned tiles_n,
                                                           unsigned total_tiles, int reverse, PsDwParams dw) {
    constexpr int BM = 32 * MB;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int NPIECE = (BM + BN) / 8;          // 1-KiB DMA pieces per K-step (8 rows each): 52 | 48
    constexpr int PPW = NPIECE / 4;                // per loader wave: 13 | 12
    static_assert(NPIECE % 4 == 0, "pieces divide over the four loader waves");
    constexpr int E_OFF = 3 * STAGE;               // epilogue constants, by tile parity: [descale piece 1 KiB | shift piece 1 KiB] x 2
    static_assert(E_OFF + 4096 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[E_OFF + 4096];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT = K / 32;
    if (blockIdx.x >= total_tiles) return;
    const unsigned ntile = (total_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const unsigned nsteps = ntile * KT;
    const unsigned rowbytes = (unsigned)K * 4u;

    const bool tn_pow2 = (tiles_n & (tiles_n - 1u)) == 0u;
    const int tn_shift = __builtin_ctz(tiles_n);
    auto tile_origin = [&](unsigned i, long long& mm0, int& nn0) {
        const unsigned lt = xcd_remap_dir(blockIdx.x + (i < ntile ? i : ntile - 1) * gridDim.x, total_tiles, reverse);
        const unsigned tm = tn_pow2 ? lt >> tn_shift : lt / tiles_n;
        mm0 = (long long)tm * (DWM != 0 ? dw.tile_rows : BM);
        nn0 = (lt - tm * tiles_n) * BN;
    };
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    // 1: any map that divides the tile, stride 1 (borders by validity masks); 2: 12 x 12 maps, stride 1, zero-bordered chunk buffer;
    // 3: 12 x 12 maps, depthwise stride 2 (TF SAME on an even map: no top / left padding) -> 6 x 6, same buffer
    // 4: no depthwise -- the global average pool behind the GEMM (the network's last pointwise layer): the tile's maps are summed
    //    from the chunk buffer and only [maps][128 channels] means leave the kernel
    constexpr bool DW = DWM != 0;
    constexpr bool BORDERED = DWM == 2 || DWM == 3;
    // SWAP (round 3): in the depthwise / pool variants the MFMA operands trade places (activations first), so a lane holds ONE
    // channel of FOUR consecutive pixels instead of four channels of one pixel.  Parking the tile in the chunk buffer is then 16
    // consecutive dwords per pixel row and instruction (ds_write_b32, conflict-free up to the harmless 2-way of the two row
    // groups) instead of a 16-byte column of 8 consecutive rows at a 128-byte pitch = 8 lanes on the same four banks (PMC r02:
    // 30 % of this kernel's LDS cycles were bank conflicts, all from these ds_write_b128; 512 -> 512 + depthwise 71 -> 61 us).
    // The four pixels of a lane lie in one image row when 4 divides the map width (bordered 12 x 12 maps; always in the
    // unbordered layouts, whose rows are the pixel indices).  14- and 7-pixel maps keep the old orientation: with a cell
    // computed per pixel (36 divisions per lane and tile) the 224 x 224 plan lost 3.5 % in a same-box A/B.
    // PADROW (round 5): 14 x 14 maps ride in the 224-row tile as 14 image rows of SIXTEEN tile rows each (two rows per image row
    // are zero operands, computed and never parked) instead of 196 consecutive pixels + 28 rows of the next map that were computed
    // and dropped anyway: the same MFMA count, but a lane's four pixels now always share an image row, so the 14 x 14 layers take the
    // SWAP orientation too (conflict-free dword parking instead of 16-byte columns: their LDS conflict share was 0.26-0.31) with no
    // per-pixel cell arithmetic at all -- tile row block = image row.  The loaders gather the rows (per-lane source offsets).
    constexpr bool PADROW = BORDERED && MW == 14 && MB == 7;
    constexpr bool SWAP = DW && (!BORDERED || MW % 4 == 0 || PADROW);

    // ---- DW epilogue, shared by both roles: the depthwise of one 32-channel chunk of the tile from the chunk buffer cb ----
    // cb layout: rows 0 .. BM-1 = the chunk's activated pointwise results [pixel][32 ch] fp32, two zero rows (taps outside the
    // map read them), then the tile's depthwise constants [12][128 ch] floats (taps 0..8, scale, shift, unused).
    // DWM == 2 / 3 (square MW x MW maps, MAPS = 32 MB / MW^2 whole ones per tile: 12 x 12 -> 2 in 288 rows, 14 x 14 -> 1 in 224,
    // 7 x 7 -> 5 in 256): the maps sit in the buffer with ZERO cells around them at a pitch of MW + 1 rows per image row -- one
    // zero cell serves as the right neighbour of x = MW - 1 and as the left neighbour of the next row's x = 0 -- and MW + 1 zero
    // rows above, between and below the maps: pixel (i, y, x) is row 1 + PITCH + IMG i + PITCH y + x, a tap (dy, dx) is that
    // row + PITCH (dy - 1) + (dx - 1), always a valid row: no masks, and the nine offsets are instruction immediates.
    constexpr int PITCH = MW + 1, IMG = (MW + 1) * PITCH;               // MW image rows + 1 border row
    constexpr int MAPS = BM / (MW * MW), MHW = MW * MW;
    constexpr int CB_ROWS = BORDERED ? 1 + PITCH + MAPS * IMG : BM + 2;
    constexpr int CB_ZROW = BM;
    constexpr int CB_CONST = CB_ROWS * ROWB;
    // 12 x 12 and 6 x 6 bordered maps: a lane takes three ADJACENT pixels of an image row (15 window reads + 9 tap reads per
    // 32-channel chunk instead of 27 + 9: stamps with the reads knocked out put 16 % of the launch in them -- 1.3 MB of 16-byte
    // LDS reads per tile at ~128 B/clk)
    constexpr bool RUN3 = DWM == 2 && BORDERED && MW % 3 == 0;
    // per item of this lane: bit t = tap t lies inside the map.  Recomputed at every tile's epilogue from an OPAQUE copy of the lane
    // id: as loop invariants these three registers (and what hipcc hoisted with them) lived through the K loop and spilled there.
    unsigned nb_mask[3] = {0, 0, 0};   // (DWM == 2: the byte offset of the item's tap (-1, -1) in the chunk buffer instead)
    auto dw_masks = [&]() __attribute__((always_inline)) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if constexpr (DWM == 4) return;
        if constexpr (RUN3) {          // one run of three adjacent pixels of an image row per lane: the offset of its first pixel's tap (-1, -1)
            const int P = 3 * ((ln >> 3) + 8 * wave);
            const int Pc = P < MAPS * MHW ? P : MAPS * MHW - 3;
            const int img = Pc / MHW, pl = Pc - MHW * img, yy = pl / MW, xx = pl - MW * yy;
            nb_mask[0] = (unsigned)((IMG * img + PITCH * yy + xx) * ROWB + 16 * (ln & 7)) | (P < MAPS * MHW ? 0u : 0x80000000u);
            return;
        }
        if constexpr (DWM == 2) {
#pragma unroll
            for (int k3 = 0; k3 < 3; ++k3) {
                const int P = (ln >> 3) + 8 * wave + 96 * k3;
                const int Pc = P < MAPS * MHW ? P : MAPS * MHW - 1;            // (lanes past the tile's maps: a duplicate, never stored)
                const int img = Pc / MHW, pl = Pc - MHW * img, yy = pl / MW, xx = pl - MW * yy;
                nb_mask[k3] = (unsigned)((IMG * img + PITCH * yy + xx) * ROWB + 16 * (ln & 7)) | (P < MAPS * MHW ? 0u : 0x80000000u);
            }
            return;
        }
        if constexpr (DWM == 3) {      // one item: output pixel P of the tile's MAPS x (MW / 2)^2; its window starts at input (2 oy, 2 ox)
            constexpr int OW2 = MW / 2, OHW = OW2 * OW2;
            const int P = min((ln >> 3) + 8 * wave, MAPS * OHW - 1);
            const int img = P / OHW, pl = P - OHW * img, oy = pl / OW2, ox = pl - OW2 * oy;
            nb_mask[0] = (unsigned)((1 + PITCH + IMG * img + PITCH * 2 * oy + 2 * ox) * ROWB + 16 * (ln & 7));
            return;
        }
        const int H = dw.HW / dw.W;
#pragma unroll
        for (int k3 = 0; k3 < 3; ++k3) {
            const int P = (ln >> 3) + 8 * wave + 96 * k3;
            const int pl = P % dw.HW, yy = pl / dw.W, xx = pl - yy * dw.W;
            unsigned mk = P < dw.tile_rows ? 1u << 16 : 0u;      // bit 16: the pixel belongs to this tile (else: computed on zeros, not stored)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ny = yy + t / 3 - 1, nx = xx + t % 3 - 1;
                mk |= (P < dw.tile_rows && (unsigned)ny < (unsigned)H && (unsigned)nx < (unsigned)dw.W ? 1u : 0u) << t;
            }
            nb_mask[k3] = mk;
        }
    };
    auto dw_chunk = [&](int c, const unsigned char* cb, long long tm0, int tn0) __attribute__((always_inline)) {
        if constexpr (DWM == 4) {
            // global average pool of the chunk: wave i < (maps per tile) sums map i; lane = (channel quad, one of 8 row groups), the groups
            // are folded with three xor-shuffles, group 0 stores the four means
            const int maps = dw.tile_rows / dw.HW;
            if (wave < maps) {
                const int q = lane & 7, part = lane >> 3;
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
                for (int r = part; r < dw.HW; r += 8) {
                    const f32x4 v = *(const f32x4*)(cb + (wave * dw.HW + r) * ROWB + 16 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sum[e] += v[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sum[e] += __shfl_xor(sum[e], 8);
                    sum[e] += __shfl_xor(sum[e], 16);
                    sum[e] += __shfl_xor(sum[e], 32);
                    sum[e] *= dw.clamp_hi;                  // (this mode: 1 / HW)
                }
                const long long map0 = tm0 / dw.HW, nmaps = M / dw.HW;
                const __amdgpu_buffer_rsrc_t ro = make_rsrc((char*)dw.ys + (map0 * Cout + tn0) * 4ll, ((nmaps - map0) * Cout - tn0) * 4ll);
                bstore16_welded(sum, ro, part == 0 ? (unsigned)wave * (unsigned)Cout * 4u + (unsigned)(32 * c + 4 * q) * 4u : 0x80000000u, 0u);
            }
            return;
        }
        const int q = lane & 7;
        const unsigned char* cp = cb + CB_CONST + (32 * c + 4 * q) * 4;
        f32x4 tap[9];          // (RUN3 only: one row of taps live at a time)
        // split rows of the output: pixel m, 32-channel group -> 128 B; this lane stores the 16-byte unit q/2 of the hi half
        // (even quad) or of the lo half (odd quad) after swapping one 8-byte half with its neighbour lane (dwconv.hip)
        // (stride 2: a quarter of the pixels -- the tile's 288 input pixels are output pixels tm0 / 4 .. + 71)
        constexpr int OSH = DWM == 3 ? 2 : 0;
        const __amdgpu_buffer_rsrc_t ro = make_rsrc((char*)dw.ys + ((tm0 >> OSH) * Cout + tn0) * 4ll, (((M - tm0) >> OSH) * Cout - tn0) * 4ll);
        const bool odd = q & 1;
        const unsigned unit = odd ? 4u + (unsigned)(q >> 1) : (unsigned)(q >> 1);
        if constexpr (RUN3) {
            const unsigned char* wb = cb + (nb_mask[0] & 0x7FFFFFFFu);
            f32x4 acc3[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) acc3[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                f32x4 v[5];
#pragma unroll
                for (int cidx = 0; cidx < 5; ++cidx) v[cidx] = *(const f32x4*)(wb + (r * PITCH + cidx) * ROWB);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) tap[3 * r + dx] = *(const f32x4*)(cp + (3 * r + dx) * 512);      // (one row of taps live at a time)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)             // (per output: taps in the order 0..8 of the one-pixel form -- same bits)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc3[j][e] = fmaf(v[j + dx][e], tap[3 * r + dx][e], acc3[j][e]);
            }
            const f32x4 sc = *(const f32x4*)(cp + 9 * 512), sh = *(const f32x4*)(cp + 10 * 512);
            const int P0 = 3 * ((lane >> 3) + 8 * wave);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fminf(fmaxf(fmaf(acc3[j][e], sc[e], sh[e]), 0.f), dw.clamp_hi);
                const f16x4 hi = __builtin_convertvector(o, f16x4);
                const f16x4 lo = __builtin_convertvector(o - __builtin_convertvector(hi, f32x4), f16x4);
                const u32x2 hb = __builtin_bit_cast(u32x2, hi), lb = __builtin_bit_cast(u32x2, lo);
                const u32x2 send = odd ? hb : lb;
                u32x2 recv;
                recv.x = (unsigned)__builtin_amdgcn_mov_dpp((int)send.x, 0xB1, 0xF, 0xF, true);
                recv.y = (unsigned)__builtin_amdgcn_mov_dpp((int)send.y, 0xB1, 0xF, 0xF, true);
                const u32x4 out = odd ? u32x4{recv.x, recv.y, lb.x, lb.y} : u32x4{hb.x, hb.y, recv.x, recv.y};
                const unsigned ovoff = (nb_mask[0] >> 31) ? 0x80000000u : (unsigned)(P0 + j) * (unsigned)Cout * 4u + (unsigned)c * 128u + 16u * unit;
                PS_EPI_STORE(__builtin_bit_cast(f32x4, out), ro, ovoff);
            }
            return;
        }
        // tap-outer order: ONE tap vector live at a time and NK accumulators (with all nine taps loaded up front -- 36 registers --
        // the masked variant spilled 27 registers, and the reloads sat in the MFMA waves' K loop); per output the products are
        // still added in tap order 0..8: same bits
        constexpr int NK = DWM == 3 ? 1 : 3;
        f32x4 a[NK];
#pragma unroll
        for (int k3 = 0; k3 < NK; ++k3) a[k3] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f32x4 tp = *(const f32x4*)(cp + t * 512);
#pragma unroll
            for (int k3 = 0; k3 < NK; ++k3) {
                const int P = (lane >> 3) + 8 * wave + 96 * k3;
                f32x4 v;
                if constexpr (BORDERED) {
                    v = *(const f32x4*)(cb + (nb_mask[k3] & 0x7FFFFFFFu) + ((t / 3) * PITCH + t % 3) * ROWB);
                } else {
                    const int row = ((nb_mask[k3] >> t) & 1u) ? P + (t / 3 - 1) * dw.W + (t % 3 - 1) : CB_ZROW;
                    v = *(const f32x4*)(cb + row * ROWB + 16 * q);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) a[k3][e] = fmaf(v[e], tp[e], a[k3][e]);
            }
        }
        const f32x4 sc = *(const f32x4*)(cp + 9 * 512), sh = *(const f32x4*)(cp + 10 * 512);
#pragma unroll
        for (int k3 = 0; k3 < NK; ++k3) {
            const int P = (lane >> 3) + 8 * wave + 96 * k3;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fminf(fmaxf(fmaf(a[k3][e], sc[e], sh[e]), 0.f), dw.clamp_hi);
            const f16x4 hi = __builtin_convertvector(o, f16x4);
            const f16x4 lo = __builtin_convertvector(o - __builtin_convertvector(hi, f32x4), f16x4);
            const u32x2 hb = __builtin_bit_cast(u32x2, hi), lb = __builtin_bit_cast(u32x2, lo);
            const u32x2 send = odd ? hb : lb;
            u32x2 recv;
            recv.x = (unsigned)__builtin_amdgcn_mov_dpp((int)send.x, 0xB1, 0xF, 0xF, true);
            recv.y = (unsigned)__builtin_amdgcn_mov_dpp((int)send.y, 0xB1, 0xF, 0xF, true);
            const u32x4 out = odd ? u32x4{recv.x, recv.y, lb.x, lb.y} : u32x4{hb.x, hb.y, recv.x, recv.y};
            // (stride 2: lanes past the tile's 72 output pixels computed a duplicate of pixel 71; an out-of-range offset drops their store)
            const bool drop = (DWM == 3 && P >= MAPS * (MW / 2) * (MW / 2)) || (DWM == 1 && !(nb_mask[k3] >> 16)) || (DWM == 2 && (nb_mask[k3] >> 31));
            const unsigned ovoff = drop ? 0x80000000u : (unsigned)P * (unsigned)Cout * 4u + (unsigned)c * 128u + 16u * unit;
            PS_EPI_STORE(__builtin_bit_cast(f32x4, out), ro, ovoff);
        }
    };

    if (wave >= 8) {
        // =================================== loader waves 8..11: all the vector-memory reads ===================================
        // (their issue stalls -- 100+ cycles per piece with the queues busy -- never hold up an MFMA wave's instruction stream)
        const int lw = wave - 8;
        unsigned pv[PPW];          // per-lane byte offset of the piece's 16 B inside the A (or B) tile, K-step 0
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int p = lw * PPW + j;
            const int r = (p < BM / 8 ? p * 8 : (p - BM / 8) * 8) + (lane >> 3);     // row inside its own tile (A or B)
            pv[j] = (unsigned)r * rowbytes + 16u * (unsigned)((lane & 7) ^ swz_key(r));
            if (PADROW && p < BM / 8)          // tile row r = image row r >> 4, column r & 15: pixel 14 (r >> 4) + (r & 15) of the map, columns 14, 15 zero
                pv[j] = (r & 15) < MW ? (unsigned)((r >> 4) * MW + (r & 15)) * rowbytes + 16u * (unsigned)((lane & 7) ^ swz_key(r)) : 0x80000000u;
        }
        auto piece = [&](const __amdgpu_buffer_rsrc_t& r, unsigned lds_addr, unsigned voff, unsigned soff) {
            // issued from asm: hipcc serialises builtin LDS-DMA against every later ds_read (DESIGN.md lesson 15b)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff),
                         "s"(r), "s"(__builtin_amdgcn_readfirstlane(soff))
                         : "memory", "m0");
        };
        __amdgpu_buffer_rsrc_t ra_rsrc, rb_rsrc;
        unsigned pf_i = 0, pf_step = 0;
        int pf_kt = 0, pf_n0 = 0;
        auto setup_rsrc = [&](unsigned i) {
            long long mm0;
            tile_origin(i, mm0, pf_n0);
            // rows beyond M / Cout fall outside the resource: the DMA writes ZEROS for them (tail tiles cost no branches)
            ra_rsrc = make_rsrc((const char*)xs + mm0 * (long long)rowbytes, (M - mm0) * (long long)rowbytes);
            rb_rsrc = make_rsrc((const char*)wsplit + (long long)pf_n0 * rowbytes, (long long)(Cout - pf_n0) * rowbytes);
        };
        auto issue_step = [&]() {      // DMA of the prefetch cursor's step into ring slot pf_step % 3; past the end it re-reads the last tile
            const unsigned base = lds0 + (pf_step % 3u) * STAGE;
            const unsigned so = (unsigned)pf_kt * 128u;
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                const int p = lw * PPW + j;       // wave-uniform: the resource is picked with scalar selects
#ifdef HSEFR_PS_KO      // knock-out builds (timing only, results WRONG; HSEFR_EXTRA_FLAGS=-DHSEFR_PS_KO=<bits>): 1 = the activation pieces of every
                        // tile whose channel origin is not 0 move no bytes (out-of-range source; the piece is still issued), 2 = the weight
                        // pieces likewise, 4 = no epilogue stores.  Compile-time on purpose: the same switch as a run-time flag made the
                        // loaders' issue loop 45 % slower (58 -> 85 us per layer) -- it must stay branch-free (lesson 19)
                const bool dead = p < BM / 8 ? (((HSEFR_PS_KO) & 1) && pf_n0 != 0) : ((HSEFR_PS_KO) & 2) != 0;
                piece(p < BM / 8 ? ra_rsrc : rb_rsrc, base + p * 1024, dead ? 0x80000000u : pv[j], dead ? 0u : so);
#else
                piece(p < BM / 8 ? ra_rsrc : rb_rsrc, base + p * 1024, pv[j], so);
#endif
            }
            ++pf_step;
            if (++pf_kt == KT) {
                pf_kt = 0;
                setup_rsrc(++pf_i);
            }
        };
        setup_rsrc(0);
        int e_n0 = pf_n0;          // output-channel origin of the tile the MFMA waves are on
        unsigned ci = 0;
        int ckt = 0;
        issue_step();
        issue_step();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        __syncthreads();
        PS_STAMP_DECL;
        for (unsigned g = 0; g < nsteps; ++g) {
            if (ckt == 0 && lw == 3) {
                // The tile's epilogue constants travel by LDS-DMA too: descale[n0 .. n0+127] as lanes 0-31 of one piece,
                // shift[..] as lanes 32-63 of a second one; the other half of each piece is out of range (zeros, unused).
                // Issued AHEAD of this step's pieces, so the counted wait below covers them.  Two copies by tile parity: the
                // MFMA waves may still be in the previous tile's epilogue (there is no barrier between tiles).
                const __amdgpu_buffer_rsrc_t rd = make_rsrc(descale + e_n0, 512), rs = make_rsrc(shift + e_n0, 512);
                const unsigned eb = lds0 + E_OFF + (ci & 1u) * 2048u;
                piece(rd, eb, 16u * lane, 0u);
                piece(rs, eb + 1024, 16u * (unsigned)(lane - 32), 0u);      // lanes 0-31 wrap to ~4 G: out of range
            }
            issue_step();                                   // step g + 2 into slot (g + 2) % 3 (released at the last barrier)
            PS_STAMP(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");   // step g + 1 (and the constants) have landed: vmcnt retires in order
            PS_STAMP(1);
            __syncthreads();
            PS_STAMP(2);
            if (++ckt == KT) {
                ckt = 0;
                long long mm0;
                if constexpr (DW) {
                    // the tile's depthwise runs on all twelve waves out of the stage the last step released
                    int tn0;
                    tile_origin(ci, mm0, tn0);
                    unsigned char* cb = smem + (unsigned)(((ci + 1u) * (unsigned)KT - 1u) % 3u) * STAGE;
                    if (DWM != 4 && lw == 3) {
                        const __amdgpu_buffer_rsrc_t rc = make_rsrc(dw.dwc, 11ll * Cout * 4);
                        const unsigned cbl = lds0 + (unsigned)(cb - smem) + CB_CONST;
#pragma unroll
                        for (int j = 0; j < 6; ++j)      // piece j = constant rows 2 j (lanes 0-31) and 2 j + 1 (lanes 32-63; row 11: out of range)
                            piece(rc, cbl + j * 1024, (unsigned)(2 * j + (lane >> 5)) * (unsigned)Cout * 4u + (unsigned)tn0 * 4u + 16u * (unsigned)(lane & 31), 0u);
                    }
                    if constexpr (BORDERED) {
                        // the zero cells of the bordered layout (the stage still holds the last step's operands): every row that is
                        // not a pixel -- row 0, the border rows, cell 12 of every image row
                        for (int r = (wave - 8) * 8 + (lane >> 3); r < CB_ROWS; r += 32) {
                            const int rr = r - 1 - PITCH, i2 = rr >= 0 ? rr / IMG : 0, r2 = rr - IMG * i2;
                            const bool pixel = rr >= 0 && r2 < MW * PITCH && r2 % PITCH != MW;
                            if (!pixel) *(f32x4*)(cb + r * ROWB + 16 * (lane & 7)) = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    } else if constexpr (DWM == 1) {
                        if (lw == 2 && lane < 16) *(f32x4*)(cb + CB_ZROW * ROWB + 16 * lane) = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    dw_masks();
#pragma unroll 1
                    for (int c = 0; c < 4; ++c) {
                        PS_STAMP(3);                        // (diagnostic builds: 3 = the tile's other work, 4 = barrier waits, 5 = the depthwise)
                        __syncthreads();
                        PS_STAMP(4);
                        dw_chunk(c, cb, mm0, tn0);
                        PS_STAMP(5);
                        __syncthreads();
                        PS_STAMP(4);
                    }
                    tile_origin(++ci, mm0, e_n0);
                } else {
                    tile_origin(++ci, mm0, e_n0);
                    __syncthreads();                            // pause while the MFMA waves store the tile (shared path to L2)
                }
                PS_STAMP(3);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the run-ahead pieces must not outlive the workgroup's LDS
        PS_STAMP_FLUSH;
        return;
    }

    // =================================== MFMA waves 0..7: wave tile rows [wm * 16 MB, +16 MB) x columns [wn * 32, +32) ===================================
    // (wm from the wave's low bit: the two waves that park a 32-channel chunk in the depthwise epilogue -- (0, wn) and (1, wn) -- are then
    // NEIGHBOURS in the workgroup, so they sit on different SIMDs and on both halves of the LDS store path; as waves wn and wn + 4 they
    // shared one SIMD's issue and one half's store rate while ten waves waited at the barrier)
    const int wm = wave & 1, wn = wave >> 1;
    const int l16 = lane & 15, lq = lane >> 4;
    // fragment byte offsets inside a stage: row (base + l16), logical chunk lq (hi) / 4 + lq (lo); + 16-row block strides
    const int arow = wm * 16 * MB + l16, brow = BM + wn * 32 + l16;
    const int a_hi = arow * ROWB + 16 * (lq ^ swz_key(arow)), a_lo = arow * ROWB + 16 * ((4 + lq) ^ swz_key(arow));
    const int b_hi = brow * ROWB + 16 * (lq ^ swz_key(brow)), b_lo = brow * ROWB + 16 * ((4 + lq) ^ swz_key(brow));
    // (a block of 16 rows further down keeps the key: key depends on row & 15 only)
    f32x4 acc[MB][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // epilogue geometry: row l16 of each 16-row block, channels 4 lq .. 4 lq + 3 of each 16-channel block
    const unsigned yvoff = ((unsigned)(wm * 16 * MB + l16) * (unsigned)Cout + (unsigned)(wn * 32 + 4 * lq)) * 4u;

    long long m0;
    int n0;
    unsigned ci = 0;
    int ckt = 0;
    tile_origin(0, m0, n0);
    __syncthreads();                                        // (the loaders' prologue barrier: step 0 has landed)
    PS_STAMP_DECL;

    // The MFMAs of a step's LAST HB row blocks are held back until after the step barrier: their fragments are already in
    // registers, so the matrix pipe has work while the barrier resolves and the next step's first fragments come out of LDS
    // (stamps of the first version: 14 % of the K loop with both waves of a SIMD waiting).  A tile's first step finds
    // zero fragments there (adding +0 changes nothing); its last step runs its own held-back blocks before the epilogue.
    // Per accumulator the products are still added in step order: same bits as without the hold-back.
    constexpr int HB = 2;
    const f16x8 fzero = {0, 0, 0, 0, 0, 0, 0, 0};
    f16x8 h_ah[HB], h_al[HB], h_bh0 = fzero, h_bl0 = fzero, h_bh1 = fzero, h_bl1 = fzero;
#pragma unroll
    for (int i = 0; i < HB; ++i) h_ah[i] = h_al[i] = fzero;
    auto mfma_block = [&](int mb, const f16x8& xh, const f16x8& xl, const f16x8& wh0, const f16x8& wl0, const f16x8& wh1, const f16x8& wl1) {
        // per accumulator the products keep the order (wh*al, wl*ah, wh*ah) of pwconv_f16s.hip
        if constexpr (SWAP) {      // lane (l16, lq) -> channel l16 of the block, pixels 4 lq .. 4 lq + 3
            acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh0, acc[mb][0], 0, 0, 0);
            acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh1, acc[mb][1], 0, 0, 0);
            acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl0, acc[mb][0], 0, 0, 0);
            acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl1, acc[mb][1], 0, 0, 0);
            acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh0, acc[mb][0], 0, 0, 0);
            acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh1, acc[mb][1], 0, 0, 0);
            return;
        }
        acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, xl, acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, xl, acc[mb][1], 0, 0, 0);
        acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl0, xh, acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl1, xh, acc[mb][1], 0, 0, 0);
        acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh0, xh, acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh1, xh, acc[mb][1], 0, 0, 0);
    };

    for (unsigned g = 0; g < nsteps; ++g) {
        const unsigned char* stg = smem + (g % 3u) * STAGE;
        const f16x8 bh0 = *(const f16x8*)(stg + b_hi), bl0 = *(const f16x8*)(stg + b_lo);
        const f16x8 bh1 = *(const f16x8*)(stg + b_hi + 16 * ROWB), bl1 = *(const f16x8*)(stg + b_lo + 16 * ROWB);
        f16x8 ah[MB], al[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            ah[mb] = *(const f16x8*)(stg + a_hi + mb * 16 * ROWB);
            al[mb] = *(const f16x8*)(stg + a_lo + mb * 16 * ROWB);
        }
        // (1) the previous step's held-back blocks, (2) this step's blocks 0 .. MB - HB - 1
#pragma unroll
        for (int i = 0; i < HB; ++i) mfma_block(MB - HB + i, h_ah[i], h_al[i], h_bh0, h_bl0, h_bh1, h_bl1);
#pragma unroll
        for (int mb = 0; mb < MB - HB; ++mb) mfma_block(mb, ah[mb], al[mb], bh0, bl0, bh1, bl1);
        // schedule: weight fragments + PRE row blocks of activation fragments up front, then per group of six MFMAs the two
        // reads of one more row block in their shadow; the first HB groups need none of this step's reads
        constexpr int PRE = 2;
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * PRE, 0);
#pragma unroll
        for (int grp = 0; grp < MB; ++grp) {
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            if (grp + PRE < MB) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
#pragma unroll
        for (int i = 0; i < HB; ++i) { h_ah[i] = ah[MB - HB + i]; h_al[i] = al[MB - HB + i]; }
        h_bh0 = bh0; h_bl0 = bl0; h_bh1 = bh1; h_bl1 = bl1;
        PS_STAMP(0);
        __syncthreads();                                // step g + 1 has landed (the loaders waited for it); slot g % 3 is released
        PS_STAMP(1);
        if (++ckt == KT) {
#pragma unroll
            for (int i = 0; i < HB; ++i) mfma_block(MB - HB + i, h_ah[i], h_al[i], h_bh0, h_bl0, h_bh1, h_bl1);
#pragma unroll
            for (int i = 0; i < HB; ++i) h_ah[i] = h_al[i] = fzero;
            h_bh0 = h_bl0 = h_bh1 = h_bl1 = fzero;
            // lane (l16, lq) holds, per 16 x 16 block, channels 4 lq .. 4 lq + 3 of row l16 (operands swapped): 16 bytes.
            // Stored as they are: one instruction covers 16 rows x 64 contiguous bytes, the block of the other 16 channels
            // completes the 128-byte lines right behind it (L2 merges the halves).  The store path of a CU (64 B/clk to L2)
            // is shared with the loaders' DMA: they pause at the barrier below until the tile is out (stamps: 2300 cycles
            // of stores alone became 6500-7200 when both ran together, all of it MFMA-wave stall).
            f32x4 e_ds[2], e_sh[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                e_ds[nb] = *(const f32x4*)(smem + E_OFF + (ci & 1u) * 2048 + (wn * 32 + 16 * nb + 4 * lq) * 4);
                e_sh[nb] = *(const f32x4*)(smem + E_OFF + (ci & 1u) * 2048 + 1024 + 512 + (wn * 32 + 16 * nb + 4 * lq) * 4);
            }
            if constexpr (DW) {
                // activated results in place, then chunk by chunk through LDS into the depthwise (all twelve waves)
                if constexpr (SWAP) {      // one channel per lane: 16 nb + l16 of the wave's 32
                    float s_ds[2], s_sh[2];
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        s_ds[nb] = *(const float*)(smem + E_OFF + (ci & 1u) * 2048 + (wn * 32 + 16 * nb + l16) * 4);
                        s_sh[nb] = *(const float*)(smem + E_OFF + (ci & 1u) * 2048 + 1024 + 512 + (wn * 32 + 16 * nb + l16) * 4);
                    }
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[mb][nb][e] = apply_act<ACT>(fmaf(acc[mb][nb][e], s_ds[nb], s_sh[nb]));
                } else {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[mb][nb][e] = apply_act<ACT>(fmaf(acc[mb][nb][e], e_ds[nb][e], e_sh[nb][e]));
                }
                unsigned char* cb = smem + (g % 3u) * STAGE;
                dw_masks();
#pragma unroll 1
                for (int c = 0; c < 4; ++c) {
                    if (SWAP && wn == c) {
                        // pixels 16 mb + 4 lq + (0..3) of the wave's half: one image row (4 | MW), so their chunk-buffer rows are consecutive
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) {
                            const int p0 = wm * 16 * MB + 16 * mb + 4 * lq;
                            if constexpr (PADROW) {      // image row wm MB + mb, columns 4 lq .. 4 lq + 3 (columns 14, 15 do not exist)
                                const int row = 1 + PITCH + PITCH * (wm * MB + mb) + 4 * lq;
#pragma unroll
                                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (4 * lq + e < MW) *(float*)(cb + (row + e) * ROWB + (16 * nb + l16) * 4) = acc[mb][nb][e];
                            } else if constexpr (BORDERED && MW == 12 && MB == 9) {
                                // wave half = map (288 rows = two 144-pixel maps), t = 4 mb + lq = the pixel quad inside the map: image row t / 3,
                                // column 4 (t % 3) -> chunk-buffer row 1 + PITCH + IMG wm + 13 (t / 3) + 4 (t % 3) = ... + 4 t + t / 3, and
                                // t / 3 = (4 mb) / 3 + ((4 mb) % 3 + lq) / 3: a compile-time part and one of three per-lane constants
                                const int row = 1 + PITCH + IMG * wm + 16 * mb + 4 * lq + (4 * mb) / 3 + ((4 * mb) % 3 + lq) / 3;
#pragma unroll
                                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) *(float*)(cb + (row + e) * ROWB + (16 * nb + l16) * 4) = acc[mb][nb][e];
                            } else {
                                int row = p0;
                                if constexpr (BORDERED) {
                                    const int pt = row < MAPS * MHW ? row : MAPS * MHW - 4;
                                    const int img = pt / MHW, pl = pt - MHW * img, yy = pl / MW;
                                    row = row < MAPS * MHW ? 1 + PITCH + IMG * img + PITCH * yy + (pl - MW * yy) : -1;
                                }
                                if (row >= 0) {
#pragma unroll
                                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                                        for (int e = 0; e < 4; ++e) *(float*)(cb + (row + e) * ROWB + (16 * nb + l16) * 4) = acc[mb][nb][e];
                                }
                            }
                        }
                    }
                    if (!SWAP && wn == c) {
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                            for (int nb = 0; nb < 2; ++nb)
                            {
                                int row = wm * 16 * MB + 16 * mb + l16;
                                if constexpr (BORDERED) {
                                    const int pt = row < MAPS * MHW ? row : MAPS * MHW - 1;      // (rows past the maps: parked on the last pixel's cell... see below)
                                    const int img = pt / MHW, pl = pt - MHW * img, yy = pl / MW;
                                    row = row < MAPS * MHW ? 1 + PITCH + IMG * img + PITCH * yy + (pl - MW * yy) : -1;
                                }
                                if (row >= 0) *(f32x4*)(cb + row * ROWB + (16 * nb + 4 * lq) * 4) = acc[mb][nb];      // (tile rows past its whole maps are not parked)
                            }
                    }
                    PS_STAMP(4);                        // (diagnostic builds: 4 = parking, 5 = barrier waits, 2 = the depthwise itself)
                    __syncthreads();
                    PS_STAMP(5);
                    dw_chunk(c, cb, m0, n0);
                    PS_STAMP(2);
                    __syncthreads();                    // (the last one lets the loaders go on)
                    PS_STAMP(5);
                }
                PS_STAMP(2);
            } else {
                const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + m0 * Cout + n0, ((M - m0) * Cout - n0) * 4ll);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = apply_act<ACT>(fmaf(acc[mb][nb][e], e_ds[nb][e], e_sh[nb][e]));
                        // rows beyond M fall outside the resource and are dropped by the hardware
                        // (asm store with its wait state welded on: every vector-memory wait in this kernel is explicit)
                        bstore16_welded(o, ry, yvoff + 64u * nb, __builtin_amdgcn_readfirstlane((unsigned)(16 * mb) * (unsigned)Cout * 4u));
                    }
            }
            zero_acc();
            ckt = 0;
            tile_origin(++ci, m0, n0);
            if constexpr (!DW) {
                PS_STAMP(2);
                __syncthreads();                            // lets the loaders go on
            }
            PS_STAMP(3);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // stores issued from asm: drained before the wave ends
    PS_STAMP_FLUSH;
}

// Tile height: 32 * MB rows with MB = 9 or 8, whichever costs fewer (rounds of 256 workgroups) x (rows per tile).
int choose_mb(long long m, int cout, int forced) {
    if (forced == 8 || forced == 9) return forced;
    int best = 9;
    long long best_cost = -1;
    for (int mb = 9; mb >= 8; --mb) {
        const long long tiles = ((m + 32 * mb - 1) / (32 * mb)) * (cout / BN);
        const long long cost = ((tiles + 255) / 256) * mb;
        if (best_cost < 0 || cost < best_cost) { best = mb; best_cost = cost; }
    }
    return best;
}

HSEFR_KNOB(g_ps_mb, 0);   // dev builds: 8 | 9 = forced tile height / 32
HSEFR_KNOB(g_ps_grid, 0); // dev builds: > 0 = forced number of persistent workgroups (contention experiments)
HSEFR_KNOB(g_psdw_mode, 0); // dev builds: 1 = the masked depthwise epilogue on 12 x 12 maps too (A/B against the bordered one)

template <int MB>
int launch_mb(const void* xs, const void* wsplit, const float* descale, const float* shift, float* y, long long m, int k, int cout,
              int act, hipStream_t s) {
    const long long tiles_m = (m + 32 * MB - 1) / (32 * MB);
    const unsigned tiles_n = cout / BN;
    const long long total = tiles_m * tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "pwconv_presplit: too many tiles");
    unsigned g = (unsigned)(total < 256 ? total : 256);
    if (g_ps_grid > 0 && (unsigned)g_ps_grid < g) g = (unsigned)g_ps_grid;
    const PsDwParams nodw{nullptr, nullptr, 0, 0, 0.f, 32 * MB};
#define HSEFR_PS_LAUNCH(A)                                                                                                 \
    HSEFR_LAUNCH((pwconv_ps_kernel<MB, A, 0>), dim3(g), dim3(768), 0, s, xs, wsplit, descale, shift, y, m, k, cout, tiles_n, \
                       (unsigned)total, sweep_reverse(), nodw)
    if (act == HSEFR_ACT_RELU6) HSEFR_PS_LAUNCH(HSEFR_ACT_RELU6);
    else if (act == HSEFR_ACT_RELU) HSEFR_PS_LAUNCH(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_NONE) HSEFR_PS_LAUNCH(HSEFR_ACT_NONE);
    else { set_error("pwconv_presplit: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_PS_LAUNCH
    return launch_status("pwconv_presplit");
}

}  // namespace

#ifdef HSEFR_DEV
void set_ps_mb(int v) { g_ps_mb = v; }
void set_ps_grid(int v) { g_ps_grid = v; }
void set_psdw_mode(int v) { g_psdw_mode = v; }
int read_ps_stamps(void* host_out, size_t bytes) {
#ifdef HSEFR_PS_STAMPS
    HSEFR_REQUIRE(bytes <= sizeof(unsigned long long) * 256 * 12 * 8, HSEFR_ERR_INVALID, "read_ps_stamps: too many bytes");
    HSEFR_HIP_CHECK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ps_stamps), bytes));
    return HSEFR_OK;
#else
    (void)host_out; (void)bytes;
    set_error("read_ps_stamps: library built without -DHSEFR_PS_STAMPS");
    return HSEFR_ERR_UNSUPPORTED;
#endif
}
#endif

// Tile height for maps of map_hw pixels: 32 MB rows hold floor(32 MB / map_hw) whole maps; the best filled of MB = 9, 8, 7
static int dw_tile_mb(int map_hw) {
    int best = 0;
    double eff = 0;
    for (int mb = 9; mb >= 7; --mb) {
        const int maps = 32 * mb / map_hw;
        const double e = (double)(maps * map_hw) / (32 * mb);
        if (maps >= 1 && e > eff + 1e-9) { eff = e; best = mb; }
    }
    return best;
}

bool pwconv_ps_dw_supported(long long m, int k, int cout, int map_w, int map_hw, int dw_stride) {
    // 3 x 3 / SAME; stride 1 on any map of at most 288 pixels (square or not), stride 2 on 12 x 12 and 14 x 14 maps
    return pwconv_ps_supported(m, k, cout) && map_w > 0 && map_hw > 0 && map_hw % map_w == 0 && map_hw <= 288 && m % map_hw == 0 &&
           (dw_stride == 1 || (dw_stride == 2 && (map_w == 12 || map_w == 14) && map_hw == map_w * map_w));
}

template <int MB, int MODE, int MW = 12>
static int launch_psdw_relu6(const void* xs, const void* wsplit, const float* descale, const float* shift, long long m, int k, int cout,
                             PsDwParams dw, hipStream_t s) {
    const long long tiles_m = (m + dw.tile_rows - 1) / dw.tile_rows;
    const unsigned tiles_n = cout / BN;
    const long long total = tiles_m * tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "pwconv_presplit_dw: too many tiles");
    const unsigned g = (unsigned)(total < 256 ? total : 256);
    HSEFR_LAUNCH((pwconv_ps_kernel<MB, HSEFR_ACT_RELU6, MODE, MW>), dim3(g), dim3(768), 0, s, xs, wsplit, descale, shift, (float*)nullptr, m, k,
                       cout, tiles_n, (unsigned)total, sweep_reverse(), dw);
    return launch_status("pwconv_presplit_dw");
}

int launch_pwconv_ps_dw(const void* xs, const void* wsplit, const float* descale, const float* shift, const float* dwc, void* ys, long long m,
                        int k, int cout, int act, int map_w, int map_hw, int dw_stride, int out_log2, hipStream_t s) {
    HSEFR_REQUIRE(pwconv_ps_dw_supported(m, k, cout, map_w, map_hw, dw_stride), HSEFR_ERR_UNSUPPORTED,
                  "pwconv_presplit_dw: m=%lld k=%d cout=%d map %d (w %d) stride %d not covered (map <= 288 pixels; stride 2: 12x12)", m, k, cout,
                  map_hw, map_w, dw_stride);
    HSEFR_REQUIRE(dw_stride == 1 || act == HSEFR_ACT_RELU6, HSEFR_ERR_UNSUPPORTED, "pwconv_presplit_dw: the stride-2 epilogue is built for ReLU6");
    HSEFR_REQUIRE(out_log2 >= 1 && out_log2 <= 12, HSEFR_ERR_INVALID, "pwconv_presplit_dw: out_log2=%d", out_log2);
    if (m == 0) return HSEFR_OK;
    const int mb = (map_w == 12 && map_hw == 144) || act != HSEFR_ACT_RELU6 ? 9 : dw_tile_mb(map_hw);      // (14 x 14 -> 7, 7 x 7 -> 8)
    PsDwParams dw{dwc, ys, map_w, map_hw, 6.f * (float)(1 << out_log2), (32 * mb / map_hw) * map_hw};
    if (act == HSEFR_ACT_RELU6) {
        const bool square = map_hw == map_w * map_w && g_psdw_mode != 1;      // the zero-bordered chunk buffer: 12 x 12, 14 x 14, 7 x 7 maps
        if (dw_stride == 2 && map_w == 14) return launch_psdw_relu6<7, 3, 14>(xs, wsplit, descale, shift, m, k, cout, dw, s);
        if (dw_stride == 2) return launch_psdw_relu6<9, 3, 12>(xs, wsplit, descale, shift, m, k, cout, dw, s);
        if (square && map_w == 12) return launch_psdw_relu6<9, 2, 12>(xs, wsplit, descale, shift, m, k, cout, dw, s);
        if (square && map_w == 14 && mb == 7) return launch_psdw_relu6<7, 2, 14>(xs, wsplit, descale, shift, m, k, cout, dw, s);
        if (square && map_w == 7 && mb == 8) return launch_psdw_relu6<8, 2, 7>(xs, wsplit, descale, shift, m, k, cout, dw, s);
        if (mb == 7) return launch_psdw_relu6<7, 1>(xs, wsplit, descale, shift, m, k, cout, dw, s);
        if (mb == 8) return launch_psdw_relu6<8, 1>(xs, wsplit, descale, shift, m, k, cout, dw, s);
        return launch_psdw_relu6<9, 1>(xs, wsplit, descale, shift, m, k, cout, dw, s);
    }
    // other activations of the GEMM (none / ReLU): the masked epilogue at MB = 9
    constexpr int MB = 9;
    const long long tiles_m = (m + dw.tile_rows - 1) / dw.tile_rows;
    const unsigned tiles_n = cout / BN;
    const long long total = tiles_m * tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "pwconv_presplit_dw: too many tiles");
    const unsigned g = (unsigned)(total < 256 ? total : 256);
#define HSEFR_PSDW_LAUNCH(A, MODE)                                                                                               \
    HSEFR_LAUNCH((pwconv_ps_kernel<MB, A, MODE>), dim3(g), dim3(768), 0, s, xs, wsplit, descale, shift, (float*)nullptr, m, k, cout, \
                       tiles_n, (unsigned)total, sweep_reverse(), dw)
    if (act == HSEFR_ACT_RELU) HSEFR_PSDW_LAUNCH(HSEFR_ACT_RELU, 1);
    else if (act == HSEFR_ACT_NONE) HSEFR_PSDW_LAUNCH(HSEFR_ACT_NONE, 1);
    else { set_error("pwconv_presplit_dw: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_PSDW_LAUNCH
    return launch_status("pwconv_presplit_dw");
}

bool pwconv_ps_gap_supported(long long m, int k, int cout, int map_hw) {
    // whole maps in a tile, at most eight of them (one wave sums one map)
    return pwconv_ps_supported(m, k, cout) && map_hw >= 33 && map_hw <= 288 && m % map_hw == 0;
}

template <int MB>
static int launch_psgap(const void* xs, const void* wsplit, const float* descale, const float* shift, long long m, int k, int cout, int act,
                        PsDwParams dw, hipStream_t s) {
    const long long tiles_m = (m + dw.tile_rows - 1) / dw.tile_rows;
    const unsigned tiles_n = cout / BN;
    const long long total = tiles_m * tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "pwconv_presplit_gap: too many tiles");
    const unsigned g = (unsigned)(total < 256 ? total : 256);
#define HSEFR_PSGAP_LAUNCH(A)                                                                                                    \
    HSEFR_LAUNCH((pwconv_ps_kernel<MB, A, 4>), dim3(g), dim3(768), 0, s, xs, wsplit, descale, shift, (float*)nullptr, m, k, cout, \
                       tiles_n, (unsigned)total, sweep_reverse(), dw)
    if (act == HSEFR_ACT_RELU6) HSEFR_PSGAP_LAUNCH(HSEFR_ACT_RELU6);
    else if (act == HSEFR_ACT_RELU) HSEFR_PSGAP_LAUNCH(HSEFR_ACT_RELU);
    else if (act == HSEFR_ACT_NONE) HSEFR_PSGAP_LAUNCH(HSEFR_ACT_NONE);
    else { set_error("pwconv_presplit_gap: act %d", act); return HSEFR_ERR_UNSUPPORTED; }
#undef HSEFR_PSGAP_LAUNCH
    return launch_status("pwconv_presplit_gap");
}

int launch_pwconv_ps_gap(const void* xs, const void* wsplit, const float* descale, const float* shift, float* y, long long m, int k, int cout,
                         int act, int map_hw, hipStream_t s) {
    HSEFR_REQUIRE(pwconv_ps_gap_supported(m, k, cout, map_hw), HSEFR_ERR_UNSUPPORTED,
                  "pwconv_presplit_gap: m=%lld k=%d cout=%d map %d not covered (33 <= map <= 288 pixels)", m, k, cout, map_hw);
    if (m == 0) return HSEFR_OK;
    const int mb = act == HSEFR_ACT_RELU6 ? dw_tile_mb(map_hw) : 9;
    const PsDwParams dw{nullptr, y, 0, map_hw, 1.0f / (float)map_hw, (32 * mb / map_hw) * map_hw};
    if (act == HSEFR_ACT_RELU6 && mb == 7) return launch_psgap<7>(xs, wsplit, descale, shift, m, k, cout, act, dw, s);
    if (act == HSEFR_ACT_RELU6 && mb == 8) return launch_psgap<8>(xs, wsplit, descale, shift, m, k, cout, act, dw, s);
    return launch_psgap<9>(xs, wsplit, descale, shift, m, k, cout, act, dw, s);
}

bool pwconv_ps_supported(long long m, int k, int cout) {
    // byte offsets inside a tile and inside the output travel in 32 bits
    return k > 0 && k % 32 == 0 && cout > 0 && cout % BN == 0 && 320ll * k * 4 < (1ll << 31) && 320ll * cout * 4 < (1ll << 31) && m >= 0;
}

int launch_pwconv_ps(const void* xs, const void* wsplit, const float* descale, const float* shift, float* y, long long m, int k,
                     int cout, int act, hipStream_t s) {
    HSEFR_REQUIRE(pwconv_ps_supported(m, k, cout), HSEFR_ERR_UNSUPPORTED,
                  "pwconv_presplit: m=%lld k=%d cout=%d not covered (k %% 32 == 0, cout %% 128 == 0)", m, k, cout);
    if (m == 0) return HSEFR_OK;
    if (choose_mb(m, cout, g_ps_mb) == 9) return launch_mb<9>(xs, wsplit, descale, shift, y, m, k, cout, act, s);
    return launch_mb<8>(xs, wsplit, descale, shift, y, m, k, cout, act, s);
}

}  // namespace hsefr
