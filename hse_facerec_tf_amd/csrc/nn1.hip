// Post-extract identification on device: L2 row normalisation and brute-force 1-NN, gfx950.
//
// Replaces preprocessing.normalize(X, 'l2') (facerec_test.py:401) and
// KNeighborsClassifier(n_neighbors=1, p=2).fit(gallery).kneighbors(probe)
// (facerec_test.py:422 scored through classifier_tester :200-207).
//
// 1-NN = a [nq x ng x d] contraction (4582 x 4582 x 1024 for LFW, 43 GFLOP) -> fp32 MFMA:
//   dist2[q, g] = |q|^2 + |g|^2 - 2 q.g ,  arg-min over g, ties to the lowest gallery index.
// One workgroup = 32 probe rows x the whole gallery; its 4 waves take gallery tiles of 32 rows
// round-robin, each keeps a per-lane running (min, index) for its 16 accumulator rows, and the
// candidates meet once at the end (wavefront shuffles, then LDS across the 4 waves).  Row norms
// are accumulated from the very fragments that feed the MFMAs.
#include <atomic>

#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* xr = x + (size_t)row * d;
    float ss = 0.f;
    for (int j = lane; j < d; j += 64) ss = fmaf(xr[j], xr[j], ss);
    ss = wave_sum64(ss);
    float nrm = sqrtf(ss);
    if (nrm == 0.f) nrm = 1.f;  // sklearn: zero rows are left as they are
    for (int j = lane; j < d; j += 64) y[(size_t)row * d + j] = xr[j] / nrm;
}

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v < bv || (v == bv && i < bi); }

__global__ __launch_bounds__(256) void nn1_kernel(const float* __restrict__ q, const float* __restrict__ g, int nq,
                                                  int ng, int d, int* __restrict__ nn_index,
                                                  float* __restrict__ nn_dist2) {
    __shared__ float s_val[4][32];
    __shared__ int s_idx[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * 32;
    const int qrow = min(q0 + li, nq - 1);
    const float* qp = q + (size_t)qrow * d + 4 * lh;

    float best_v[16];
    int best_i[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { best_v[r] = INFINITY; best_i[r] = 0x7fffffff; }

    float qq = 0.f;  // |q_row|^2, accumulated on the first gallery tile only
    bool qq_done = false;
    const int gtiles = (ng + 31) / 32;
    for (int gt = wave; gt < gtiles; gt += 4) {
        const int gcol = gt * 32 + li;
        const int grow = min(gcol, ng - 1);
        const float* gp = g + (size_t)grow * d + 4 * lh;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float gg = 0.f, qs = 0.f;
        for (int k = 0; k < d; k += 8) {
            const f32x4 a = *(const f32x4*)(qp + k);
            const f32x4 b = *(const f32x4*)(gp + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
                gg = fmaf(b[j], b[j], gg);
                qs = fmaf(a[j], a[j], qs);
            }
        }
        gg += __shfl_xor(gg, 32);
        if (!qq_done) { qq = qs + __shfl_xor(qs, 32); qq_done = true; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;  // probe row of this accumulator register
            const float qr = __shfl(qq, rr);
            const float v = fmaxf(qr + gg - 2.f * acc[r], 0.f);
            if (gcol < ng && better(v, gcol, best_v[r], best_i[r])) { best_v[r] = v; best_i[r] = gcol; }
        }
    }
    // arg-min across the 32 gallery columns held by the lanes of each half-wave
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best_v[r], m);
            const int oi = __shfl_xor(best_i[r], m);
            if (better(ov, oi, best_v[r], best_i[r])) { best_v[r] = ov; best_i[r] = oi; }
        }
    }
    if (li == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
            s_val[wave][rr] = best_v[r];
            s_idx[wave][rr] = best_i[r];
        }
    }
    __syncthreads();
    if (threadIdx.x < 32 && q0 + threadIdx.x < nq) {
        float bv = s_val[0][threadIdx.x];
        int bi = s_idx[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (better(s_val[w][threadIdx.x], s_idx[w][threadIdx.x], bv, bi)) { bv = s_val[w][threadIdx.x]; bi = s_idx[w][threadIdx.x]; }
        nn_index[q0 + threadIdx.x] = bi;
        if (nn_dist2) nn_dist2[q0 + threadIdx.x] = bv;
    }
}

// Full Euclidean distance matrix D[i,j] = |x_i - y_j| (sklearn pairwise_distances at
// facial_clustering_test.py:396; the feature term of process_photos.py:46-51's O(N^2) Python loop),
// same MFMA contraction as nn1_kernel.  Workgroup = 64x64 outputs, wave = 32x32.
__global__ __launch_bounds__(256) void pairwise_dist_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, int m,
                                                            int d, float* __restrict__ out, int same) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int i0 = blockIdx.y * 64 + (wave >> 1) * 32, j0 = blockIdx.x * 64 + (wave & 1) * 32;
    if (i0 >= n || j0 >= m) return;
    const float* xp = x + (size_t)min(i0 + li, n - 1) * d + 4 * lh;
    const float* yp = y + (size_t)min(j0 + li, m - 1) * d + 4 * lh;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float xx = 0.f, yy = 0.f;
    for (int k = 0; k < d; k += 8) {
        const f32x4 a = *(const f32x4*)(xp + k);
        const f32x4 b = *(const f32x4*)(yp + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
            xx = fmaf(a[e], a[e], xx);
            yy = fmaf(b[e], b[e], yy);
        }
    }
    xx += __shfl_xor(xx, 32);
    yy += __shfl_xor(yy, 32);
    const int col = j0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float xr = __shfl(xx, rr);
        const int row = i0 + rr;
        float v = sqrtf(fmaxf(xr + yy - 2.f * acc[r], 0.f));
        if (same && row == col) v = 0.f;     // pairwise_distances(X) has an exact zero diagonal
        if (row < n && col < m) out[(size_t)row * m + col] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Large searches (the LFW protocol: 4582 probes x 4582 gallery rows x 1024): the contraction on the split-f16
// GEMM of pwconv_f16s.hip (three f16 MFMA products per fp32 product, error <= 3 * 2^-22 per product -- the arithmetic every
// pointwise layer of the network runs on) instead of the fp32 matrix pipe (29 TFLOP/s in nn1_kernel):
//   1. row norms |q|^2 and the largest |q| of the whole probe matrix (one device scalar: the probes are scaled by the power of
//      two that brings them into [-1, 1], which is what the GEMM's fixed activation scale 2^12 assumes);
//   2. the gallery as the GEMM's WEIGHTS: every row scaled by its own power of two into [8192, 16384), split hi / lo into
//      "split rows" [row][k / 32][hi 32 x f16 | lo 32 x f16]; descale[g] = -2 * 2^(Eq - 12 - e_g), shift[g] = |g|^2 -- so
//      the GEMM's own epilogue returns |g|^2 - 2 q.g; rows that pad the gallery to a multiple of 64 get descale 0 and a huge shift;
//   3. the GEMM -> [nq, ng_pad] fp32 (84 MB for LFW: written and read once at HBM speed);
//   4. row arg-min of max(|q|^2 + that, 0), ties to the lowest gallery index (the rule of nn1_kernel and of scikit-learn).
__global__ __launch_bounds__(256) void nn1_prep_queries_kernel(const float* __restrict__ q, int nq, int d, float* __restrict__ qq,
                                                               unsigned* __restrict__ qmax_bits) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nq) return;
    const float* r = q + (size_t)row * d;
    float ss = 0.f, mx = 0.f;
    for (int j = lane; j < d; j += 64) { const float v = r[j]; ss = fmaf(v, v, ss); mx = fmaxf(mx, fabsf(v)); }
    ss = wave_sum64(ss);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    if (lane == 0) {
        qq[row] = ss;
        atomicMax(qmax_bits, __float_as_uint(mx));          // non-negative floats order like their bit patterns (a NaN row wins: results NaN, as before)
    }
}

__device__ __forceinline__ int pow2_above(float mx) {      // e with mx * 2^-e in [0.5, 1)  (0 for mx = 0 / Inf / NaN)
    int e = 0;
    if (mx > 0.f && mx < INFINITY) (void)frexpf(mx, &e);
    return e;
}

__global__ __launch_bounds__(256) void nn1_scale_queries_kernel(const float* __restrict__ q, float* __restrict__ qs, long long n4,
                                                                const unsigned* __restrict__ qmax_bits) {
    const float sc = ldexpf(1.f, -pow2_above(__uint_as_float(*qmax_bits)));
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) ((f32x4*)qs)[i] = ((const f32x4*)q)[i] * sc;
}

__global__ __launch_bounds__(256) void nn1_split_gallery_kernel(const float* __restrict__ g, int ng, int ng_pad, int d,
                                                                unsigned short* __restrict__ wsplit, float* __restrict__ descale,
                                                                float* __restrict__ shift, const unsigned* __restrict__ qmax_bits) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= ng_pad) return;
    unsigned short* w = wsplit + (size_t)row * d * 2;
    if (row >= ng) {
        for (int j = lane; j < 2 * d; j += 64) w[j] = 0;
        if (lane == 0) { descale[row] = 0.f; shift[row] = 3.0e38f; }
        return;
    }
    const float* r = g + (size_t)row * d;
    float ss = 0.f, mx = 0.f;
    for (int j = lane; j < d; j += 64) { const float v = r[j]; ss = fmaf(v, v, ss); mx = fmaxf(mx, fabsf(v)); }
    ss = wave_sum64(ss);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    const int eg = 14 - pow2_above(mx);                      // mx * 2^eg in [8192, 16384)
    const float sc = ldexpf(1.f, eg);
    for (int j = lane; j < d; j += 64) {
        const float v = r[j] * sc;                           // exact
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const int grp = j >> 5, k = j & 31;
        w[grp * 64 + k] = __builtin_bit_cast(unsigned short, hi);
        w[grp * 64 + 32 + k] = __builtin_bit_cast(unsigned short, lo);
    }
    if (lane == 0) {
        const int eq = pow2_above(__uint_as_float(*qmax_bits));
        descale[row] = -2.f * ldexpf(1.f, eq - 12 - eg);
        shift[row] = ss;
    }
}

__global__ __launch_bounds__(256) void nn1_row_argmin_kernel(const float* __restrict__ y, int nq, int ng, int ng_pad,
                                                             const float* __restrict__ qq, int* __restrict__ nn_index,
                                                             float* __restrict__ nn_dist2) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nq) return;
    const float* r = y + (size_t)row * ng_pad;
    const float qr = qq[row];
    float bv = INFINITY;
    int bi = 0x7fffffff;
    for (int j = lane; j < ng; j += 64) {
        const float v = fmaxf(qr + r[j], 0.f);
        if (better(v, j, bv, bi)) { bv = v; bi = j; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(bv, m);
        const int oi = __shfl_xor(bi, m);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
        nn_index[row] = bi;
        if (nn_dist2) nn_dist2[row] = bv;
    }
}

}  // namespace

int launch_pairwise_dist(const float* x, const float* y, int n, int m, int d, float* out, hipStream_t s) {
    HSEFR_REQUIRE(d > 0 && d % 8 == 0, HSEFR_ERR_UNSUPPORTED, "pairwise_dist: d=%d must be a multiple of 8", d);
    HSEFR_REQUIRE(n >= 0 && m >= 0, HSEFR_ERR_INVALID, "pairwise_dist: bad shape");
    if (n == 0 || m == 0) return HSEFR_OK;
    dim3 grid((m + 63) / 64, (n + 63) / 64), block(256);
    HSEFR_LAUNCH(pairwise_dist_kernel, grid, block, 0, s, x, y, n, m, d, out, x == y ? 1 : 0);
    return launch_status("pairwise_dist");
}

int launch_l2_normalize(const float* x, float* y, int n, int d, hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && d > 0, HSEFR_ERR_INVALID, "l2_normalize: bad shape");
    if (n == 0) return HSEFR_OK;
    dim3 grid((n + 3) / 4), block(256);
    HSEFR_LAUNCH(l2_normalize_kernel, grid, block, 0, s, x, y, n, d);
    return launch_status("l2_normalize");
}

HSEFR_KNOB(g_nn1_y_mb, 256);      // dev builds: bound of the distance-matrix slice in MiB (tests force several query blocks with a small one)
#define NN1_Y_BYTES ((long long)g_nn1_y_mb << 20)
#ifdef HSEFR_DEV
void set_nn1_y_mb(int v) { g_nn1_y_mb = v > 0 ? v : 1; }
#endif

static std::atomic<long long> g_nn1_fallbacks{0};
long long nn1_fallbacks() { return g_nn1_fallbacks.load(); }

int launch_nn1(const float* q, const float* g, int nq, int ng, int d, int* nn_index, float* nn_dist2, hipStream_t s) {
    HSEFR_REQUIRE(d > 0 && d % 8 == 0, HSEFR_ERR_UNSUPPORTED, "nn1: d=%d must be a multiple of 8", d);
    HSEFR_REQUIRE(nq >= 0 && ng > 0, HSEFR_ERR_INVALID, "nn1: nq=%d ng=%d", nq, ng);
    if (nq == 0) return HSEFR_OK;
    if (d % 32 == 0 && (long long)nq * ng * d >= (1ll << 28)) {
        // the split-f16 GEMM path; its workspace (scaled probes, split gallery, a [block, ng_pad] slice of the distance matrix) is
        // stream-ordered.  The matrix is walked in QUERY BLOCKS so that it stays bounded (NN1_Y_BYTES: 100 000 x 100 000 would be
        // 40 GB in one piece -- ADVICE r4); rows are independent, so the blocking changes no bit.  If the allocation fails (the
        // caller's allocator may hold the memory) the search runs on nn1_kernel, which needs no workspace.
        const int ng_pad = (ng + 63) / 64 * 64;
        const long long y_rows_max = NN1_Y_BYTES / ((long long)ng_pad * 4);
        int qb = (int)(y_rows_max >= nq ? nq : (y_rows_max < 256 ? 256 : y_rows_max / 256 * 256));     // query rows per block
        const size_t b_qs = (size_t)nq * d * 4, b_w = (size_t)ng_pad * d * 4;
        size_t b_y = (size_t)qb * ng_pad * 4;
        const size_t b_small = ((size_t)nq + 2 * (size_t)ng_pad + 64) * 4;
        char* ws = nullptr;
        // a failed allocation is retried with HALF the query block, down to 256 rows (the slice of the distance matrix is
        // the only part of the workspace that can shrink), before the search gives up on this path
        while (hipMallocAsync((void**)&ws, b_qs + b_w + b_y + b_small, s) != hipSuccess || !ws) {
            (void)hipGetLastError();
            ws = nullptr;
            if (qb <= 256) break;
            qb = qb / 2 < 256 ? 256 : qb / 2 / 256 * 256;
            b_y = (size_t)qb * ng_pad * 4;
        }
        if (ws) {
            float* qs = (float*)ws;
            unsigned short* w = (unsigned short*)(ws + b_qs);
            float* y = (float*)(ws + b_qs + b_w);
            float* qq = (float*)(ws + b_qs + b_w + b_y);
            float* descale = qq + nq;
            float* shift = descale + ng_pad;
            unsigned* qmax = (unsigned*)(shift + ng_pad);
            int rc = HSEFR_OK;
            if (hipMemsetAsync(qmax, 0, 4, s) != hipSuccess) rc = HSEFR_ERR_HIP;
            if (rc == HSEFR_OK) {
                HSEFR_LAUNCH(nn1_prep_queries_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, q, nq, d, qq, qmax);
                HSEFR_LAUNCH(nn1_split_gallery_kernel, dim3((ng_pad + 3) / 4), dim3(256), 0, s, g, ng, ng_pad, d, w, descale, shift, qmax);
                const long long n4 = (long long)nq * d / 4;
                HSEFR_LAUNCH(nn1_scale_queries_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, q, qs, n4, qmax);
                rc = launch_status("nn1 (split-f16 preparation)");
            }
            for (int q0 = 0; q0 < nq && rc == HSEFR_OK; q0 += qb) {
                const int m = nq - q0 < qb ? nq - q0 : qb;
                rc = launch_pwconv_f16s(qs + (size_t)q0 * d, w, descale, shift, y, m, d, ng_pad, 12, HSEFR_ACT_NONE, s);
                if (rc == HSEFR_OK) {
                    HSEFR_LAUNCH(nn1_row_argmin_kernel, dim3((m + 3) / 4), dim3(256), 0, s, y, m, ng, ng_pad, qq + q0, nn_index + q0,
                                       nn_dist2 ? nn_dist2 + q0 : nullptr);
                    rc = launch_status("nn1 (row arg-min)");
                }
            }
            (void)hipFreeAsync(ws, s);
            return rc;
        }
        // no workspace even for a 256-row block: the search runs on nn1_kernel (no workspace; another summation order, far slower at
        // these sizes).  Not an error -- but COUNTED and described, so a perf cliff or a last-bit difference can be traced to it
        // (hsefr_nn1_fallbacks, hsefr_last_error_string)
        g_nn1_fallbacks.fetch_add(1);
        set_error("nn1: no stream-ordered workspace (%zu bytes) for the split-f16 search of %d x %d x %d; ran the workspace-free kernel",
                  b_qs + b_w + b_y + b_small, nq, ng, d);
    }
    dim3 grid((nq + 31) / 32), block(256);
    HSEFR_LAUNCH(nn1_kernel, grid, block, 0, s, q, g, nq, ng, d, nn_index, nn_dist2);
    return launch_status("nn1");
}

}  // namespace hsefr
