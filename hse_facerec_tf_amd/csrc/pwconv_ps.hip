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
//   * ONE persistent workgroup of 8 waves per CU, tile (32 * MB) x 128 with MB = 8 or 9 (256 or 288 rows): every layer
//     of MobileNet-192 at batch 256 has M = 9 * 2^k rows, so 288-row tiles fill whole rounds of 256 workgroups
//     (36864 x 512 -> 512 tiles = 2.0 rounds, where 128 x 128 tiles gave 2.25 rounds that cost 3); 224-pixel inputs
//     (M = 49 * 2^k) take whichever of the two needs fewer tile-rounds;
//   * three LDS stages of (BM + 128) x 128 B (156 KB for MB = 9): the DMA runs TWO K-steps ahead of the MFMAs;
//   * waves as 2 (M) x 4 (N), wave tile (16 * MB) x 32 on v_mfma_f32_16x16x32_f16 (one instruction per 32-deep K-step and
//     product; it sustains a higher clock than 32x32x16 on this chip -- DESIGN.md lesson 10), operands swapped so a lane
//     owns 4 consecutive output channels, 72 accumulator registers;
//   * epilogue through a wave-private 4 KB scratch inside the stage the tile's last step has released: 128-B-line stores.
//
// Every output element is accumulated over K in ONE fixed order by ONE wave: results are bit-identical run to run and
// independent of the grid.  (They are NOT bit-identical to pwconv_f16s.hip: a 16x16x32 MFMA sums its 32 products in a
// different internal order than two 32x32x16 steps -- same error bound, same 2e-6 test bar.)
#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int ROWB = 128;       // bytes per split row: 32 hi halves | 32 lo halves
constexpr int BN = 128;

// swizzle key of a tile row: the 16-B chunk c of row r lives at chunk position c ^ key(r) (pwconv_f16s.hip's swizzle)
__device__ __forceinline__ int swz_key(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 2); }

template <int MB, int ACT>
__global__ __launch_bounds__(512, 1) void pwconv_ps_kernel(const void* __restrict__ xs, const void* __restrict__ wsplit,
                                                           const float* __restrict__ descale, const float* __restrict__ shift,
                                                           float* __restrict__ y, long long M, int K, int Cout, unsigned tiles_n,
                                                           unsigned total_tiles, int reverse) {
    constexpr int BM = 32 * MB;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int NPIECE = (BM + BN) / 8;          // 1-KiB DMA pieces per K-step (8 rows each)
    constexpr int PPW_HI = (NPIECE + 7) / 8;       // pieces issued by waves 0 .. NPIECE % 8 - 1 (all of them if NPIECE % 8 == 0)
    constexpr int N_HI = NPIECE % 8 == 0 ? 8 : NPIECE % 8;
    constexpr int PPW_LO = NPIECE / 8;
    constexpr int E_OFF = 3 * STAGE;               // epilogue constants of the current tile: [descale piece 1 KiB | shift piece 1 KiB]
    static_assert(E_OFF + 2048 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[E_OFF + 2048];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
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
        mm0 = (long long)tm * BM;
        nn0 = (lt - tm * tiles_n) * BN;
    };

    // ---- DMA duty of this wave: pieces p = first_piece .. first_piece + npieces - 1 of every step ----------------------
    const int npieces = wave < N_HI ? PPW_HI : PPW_LO;
    const int first_piece = wave < N_HI ? wave * PPW_HI : N_HI * PPW_HI + (wave - N_HI) * PPW_LO;
    unsigned pv[PPW_HI];          // per-lane byte offset of the piece's 16 B inside the A (or B) tile, K-step 0
    {
#pragma unroll
        for (int j = 0; j < PPW_HI; ++j) {
            const int p = first_piece + (j < npieces ? j : 0);
            const int r = (p < BM / 8 ? p * 8 : (p - BM / 8) * 8) + (lane >> 3);     // row inside its own tile (A or B)
            pv[j] = (unsigned)r * rowbytes + 16u * (unsigned)((lane & 7) ^ swz_key(r));
        }
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    auto piece = [&](const __amdgpu_buffer_rsrc_t& r, unsigned lds_addr, unsigned voff, unsigned soff) {
        // issued from asm: hipcc serialises builtin LDS-DMA against every later ds_read (DESIGN.md lesson 15b)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff),
                     "s"(r), "s"(__builtin_amdgcn_readfirstlane(soff))
                     : "memory");
    };
    __amdgpu_buffer_rsrc_t ra_rsrc, rb_rsrc;
    unsigned pf_i = 0, pf_step = 0;
    int pf_kt = 0;
    auto setup_rsrc = [&](unsigned i) {
        long long mm0;
        int nn0;
        tile_origin(i, mm0, nn0);
        // rows beyond M / Cout fall outside the resource: the DMA writes ZEROS for them (tail tiles cost no branches)
        ra_rsrc = make_rsrc((const char*)xs + mm0 * (long long)rowbytes, (M - mm0) * (long long)rowbytes);
        rb_rsrc = make_rsrc((const char*)wsplit + (long long)nn0 * rowbytes, (long long)(Cout - nn0) * rowbytes);
    };
    auto issue_step = [&]() {      // DMA of the prefetch cursor's step into ring slot pf_step % 3; past the end it re-reads the last tile
        const unsigned base = lds0 + (pf_step % 3u) * STAGE;
        const unsigned so = (unsigned)pf_kt * 128u;
#pragma unroll
        for (int j = 0; j < PPW_HI; ++j) {
            if (j < npieces) {
                const int p = first_piece + j;       // wave-uniform: the resource is picked with scalar selects
                piece(p < BM / 8 ? ra_rsrc : rb_rsrc, base + p * 1024, pv[j], so);
            }
        }
        ++pf_step;
        if (++pf_kt == KT) {
            pf_kt = 0;
            setup_rsrc(++pf_i);
        }
    };
    auto wait_all_but_last_step = [&]() {      // everything older than the step issued last has landed (vmcnt retires in order)
        if (wave < N_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW_HI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW_LO) : "memory");
    };

    // ---- MFMA duty: wave tile rows [wm * 16 MB, +16 MB) x columns [wn * 32, +32) --------------------------------------
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

    // epilogue geometry: lane reads back row (lane >> 3) + 8 i of a 32-row scratch, 16-B chunk (lane & 7) = 4 channels
    const int erow = lane >> 3, ech = lane & 7;
    const unsigned yvoff = ((unsigned)(wm * 16 * MB + erow) * (unsigned)Cout + (unsigned)(wn * 32 + 4 * ech)) * 4u;

    long long m0;
    int n0;
    unsigned ci = 0;
    int ckt = 0;
    tile_origin(0, m0, n0);
    setup_rsrc(0);
    issue_step();
    issue_step();
    wait_all_but_last_step();
    __syncthreads();

    for (unsigned g = 0; g < nsteps; ++g) {
        const bool last = ckt == KT - 1;
        if (ckt == 0 && wave == 7) {
            // The tile's epilogue constants travel by LDS-DMA too (no VGPR, no compiler-side wait): descale[n0 .. n0+127] as
            // lanes 0-31 of one piece, shift[..] as lanes 32-63 of a second one; the other half of each piece is out of range
            // (zeros, unused).  Issued AHEAD of this step's pieces, so the counted wait below covers them; the previous
            // tile's epilogue is behind a barrier.
            const __amdgpu_buffer_rsrc_t rd = make_rsrc(descale + n0, 512), rs = make_rsrc(shift + n0, 512);
            piece(rd, lds0 + E_OFF, 16u * lane, 0u);
            piece(rs, lds0 + E_OFF + 1024, 16u * (unsigned)(lane - 32), 0u);      // lanes 0-31 wrap to ~4 G: out of range
        }
        issue_step();                                   // step g + 2 into slot (g + 2) % 3 (released at the last barrier)
        const unsigned char* st = smem + (g % 3u) * STAGE;
        const f16x8 bh0 = *(const f16x8*)(st + b_hi), bl0 = *(const f16x8*)(st + b_lo);
        const f16x8 bh1 = *(const f16x8*)(st + b_hi + 16 * ROWB), bl1 = *(const f16x8*)(st + b_lo + 16 * ROWB);
        f16x8 ah[MB], al[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            ah[mb] = *(const f16x8*)(st + a_hi + mb * 16 * ROWB);
            al[mb] = *(const f16x8*)(st + a_lo + mb * 16 * ROWB);
        }
        // two row blocks at a time: four independent accumulators between two MFMAs on the same one; per accumulator the
        // products keep the order (wh*al, wl*ah, wh*ah) of pwconv_f16s.hip
#pragma unroll
        for (int mb = 0; mb < MB; mb += 2) {
#pragma unroll
            for (int pdt = 0; pdt < 3; ++pdt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (mb + h < MB) {
                        const f16x8 a = pdt == 0 ? al[mb + h] : ah[mb + h];
                        acc[mb + h][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pdt == 1 ? bl0 : bh0, a, acc[mb + h][0], 0, 0, 0);
                        acc[mb + h][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pdt == 1 ? bl1 : bh1, a, acc[mb + h][1], 0, 0, 0);
                    }
                }
        }
        // schedule: the weight fragments and PRE row blocks of activation fragments up front, then per row block its six
        // MFMAs with the two reads of the block PRE further down in their shadow (LDS latency never exposed, ~40 fragment
        // registers live)
        constexpr int PRE = 4;
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * (PRE < MB ? PRE : MB), 0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            if (mb + PRE < MB) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        wait_all_but_last_step();                       // step g + 1 (and this tile's constants) have landed
        __syncthreads();                                // ... for every wave; slot g % 3 is released
        if (last) {
            // lane (l16, lq) holds, per 16 x 16 block, channels 4 lq .. 4 lq + 3 of row l16 (operands swapped).  Two row
            // blocks at a time go through a wave-private 32-row x 128 B scratch in the released slot and leave as
            // stores of 8 rows x 128 B: whole lines.
            unsigned char* scr = smem + (g % 3u) * STAGE + wave * 4096;
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + m0 * Cout + n0, ((M - m0) * Cout - n0) * 4ll);
            const f32x4 e_ds = *(const f32x4*)(smem + E_OFF + (wn * 32 + 4 * ech) * 4);
            const f32x4 e_sh = *(const f32x4*)(smem + E_OFF + 1024 + 512 + (wn * 32 + 4 * ech) * 4);
#pragma unroll
            for (int pr = 0; pr < (MB + 1) / 2; ++pr) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int mb = 2 * pr + h;
                    if (mb < MB) {
                        const int r = 16 * h + l16;
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) *(f32x4*)(scr + r * 128 + 16 * ((4 * nb + lq) ^ (r & 7))) = acc[mb][nb];
                    }
                }
                f32x4 rb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = erow + 8 * i;
                    if (2 * pr * 16 + 8 * i < 16 * MB) rb[i] = *(const f32x4*)(scr + r * 128 + 16 * (ech ^ (r & 7)));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (2 * pr * 16 + 8 * i < 16 * MB) {          // the odd last block of MB = 9 fills only half the scratch
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = apply_act<ACT>(fmaf(rb[i][e], e_ds[e], e_sh[e]));
                        // rows beyond M fall outside the resource and are dropped by the hardware
                        // (asm store with its wait state welded on: every vector-memory wait in this kernel is explicit)
                        bstore16_welded(o, ry, yvoff, __builtin_amdgcn_readfirstlane((unsigned)(32 * pr + 8 * i) * (unsigned)Cout * 4u));
                    }
                }
            }
            zero_acc();
            ckt = 0;
            ++ci;
            tile_origin(ci, m0, n0);
            __syncthreads();                            // the scratch is the slot the next iteration's DMA refills
        } else {
            ++ckt;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the run-ahead DMA pieces must not outlive the workgroup's LDS
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

template <int MB>
int launch_mb(const void* xs, const void* wsplit, const float* descale, const float* shift, float* y, long long m, int k, int cout,
              int act, hipStream_t s) {
    const long long tiles_m = (m + 32 * MB - 1) / (32 * MB);
    const unsigned tiles_n = cout / BN;
    const long long total = tiles_m * tiles_n;
    HSEFR_REQUIRE(total < (1ll << 31), HSEFR_ERR_UNSUPPORTED, "pwconv_presplit: too many tiles");
    const unsigned g = (unsigned)(total < 256 ? total : 256);
#define HSEFR_PS_LAUNCH(A)                                                                                                 \
    hipLaunchKernelGGL((pwconv_ps_kernel<MB, A>), dim3(g), dim3(512), 0, s, xs, wsplit, descale, shift, y, m, k, cout, tiles_n, \
                       (unsigned)total, sweep_reverse())
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
#endif

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
