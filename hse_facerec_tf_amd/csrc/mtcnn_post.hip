// MTCNN's box logic on the device: candidate generation from the P-Net maps, greedy NMS, box regression, squaring, the crop
// windows of the next stage and the landmark transform (generateBoundingBox / nms / bbreg / rerec / pad of
// facial_analysis.py:354-476, driven by mtcnn_detect_faces :478-604).  With the pyramid and the crops already resampled on the
// GPU (area_resize.hip) the cascade's host side shrinks to three 32-byte read-backs of box counts per frame.
//
// Parity contract: the SAME numbers as the host restatement in mtcnn.py (float64 box arithmetic, float32 scores / regressions /
// landmarks, products and sums rounded separately as NumPy does: every a + b * c below is written with __dmul_rn / __dadd_rn so
// that hipcc cannot contract it), the same greedy order: descending score, ties by ascending candidate index (for P-Net
// candidates: the row-major index of the cell in the [W', H'] map, which is the order np.nonzero yields).
//
// One workgroup of 1024 threads per call (a frame has a few hundred candidates per level): candidates are appended unordered
// with an LDS atomic, a bitonic sort of 64-bit keys (score | index) makes the order deterministic, the greedy pass walks the
// sorted list with one barrier per KEPT box.  More than MTCNN_CAP candidates in one list raise the overflow flag; the Python
// side then redoes that frame on its host path.
#include "common.h"

namespace hsefr {

namespace {

constexpr int CAP = 2048;          // candidates per list (per pyramid level; all levels' survivors; stage-2 / stage-3 inputs)
constexpr int NT = 1024;

struct PostLds {
    unsigned long long key[CAP];   // (0xFFFFFFFF - score bits) << 32 | index: ascending = score descending, index ascending
    double x1[CAP], y1[CAP], x2[CAP], y2[CAP], area[CAP];   // in SORTED order
    unsigned char alive[CAP], kept[CAP];
    int n, nkeep;
    int keep_pos[CAP];             // sorted positions of the kept boxes, in pick order
};

__device__ __forceinline__ double dfix(double v) { return trunc(v); }
__device__ __forceinline__ double mad_sep(double a, double b, double c) { return __dadd_rn(a, __dmul_rn(b, c)); }   // a + b * c, two roundings

// bitonic sort of key[0 .. np2) ascending (np2 = power of two >= n, padded with all-ones keys)
__device__ void sort_keys(PostLds& L, int np2) {
    const int tid = threadIdx.x;
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = L.key[i], b = L.key[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { L.key[i] = b; L.key[ixj] = a; }
                }
            }
            __syncthreads();
        }
}

// greedy NMS over the sorted boxes in LDS (x1..y2, area filled for positions 0 .. n-1): kept[] and keep_pos[] / nkeep
__device__ void nms_sorted(PostLds& L, int n, double thr, bool use_min) {
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += NT) { L.alive[i] = 1; L.kept[i] = 0; }
    if (tid == 0) L.nkeep = 0;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        if (!L.alive[i]) continue;                        // LDS broadcast read: uniform across the workgroup
        const double tx1 = L.x1[i], ty1 = L.y1[i], tx2 = L.x2[i], ty2 = L.y2[i], ta = L.area[i];
        for (int j = i + 1 + tid; j < n; j += NT) {
            if (!L.alive[j]) continue;
            const double iw = fmax(0.0, fmin(tx2, L.x2[j]) - fmax(tx1, L.x1[j]) + 1.0);
            const double ih = fmax(0.0, fmin(ty2, L.y2[j]) - fmax(ty1, L.y1[j]) + 1.0);
            const double inter = __dmul_rn(iw, ih);
            const double den = use_min ? fmin(ta, L.area[j]) : __dadd_rn(__dadd_rn(ta, L.area[j]), -inter);
            if (!(inter / den <= thr)) L.alive[j] = 0;
        }
        if (tid == 0) { L.kept[i] = 1; L.keep_pos[L.nkeep++] = i; }
        __syncthreads();
    }
    __syncthreads();
}

// Sort key: ascending key = descending score, and among bit-equal scores DESCENDING index -- what the reference's own
// `I = np.argsort(s)` + "pick I[-1]" does on its usual list sizes (NumPy's introsort finishes lists of <= 16 elements with a
// stable insertion sort, facial_analysis.py:398-403), made the rule for every size so the order is deterministic.
__device__ __forceinline__ unsigned long long make_key(float score, unsigned idx) {
    return ((unsigned long long)(0xFFFFFFFFu - __float_as_uint(score)) << 32) | (0xFFFFFFFFu - idx);
}
__device__ __forceinline__ unsigned key_idx(unsigned long long key) { return 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull); }
__device__ __forceinline__ int np2_of(int n) {
    int p = 2;
    while (p < n) p <<= 1;
    return p;
}

// the source window of a box clipped to the frame and where it lands in the box-sized tile (pad(), :432-466): table row
// {x1, y1, x2, y2, tx1, ty1, bw, bh} as hsefr_mtcnn_crops takes it
__device__ void crop_row(const double* b, int img_w, int img_h, int* t) {
    const int bw = (int)(b[2] - b[0] + 1.0), bh = (int)(b[3] - b[1] + 1.0);
    int x1 = (int)b[0], y1 = (int)b[1], x2 = (int)b[2], y2 = (int)b[3];
    int tx1 = 1, ty1 = 1;
    if (x2 > img_w) x2 = img_w;
    if (y2 > img_h) y2 = img_h;
    if (x1 < 1) { tx1 = 2 - x1; x1 = 1; }
    if (y1 < 1) { ty1 = 2 - y1; y1 = 1; }
    t[0] = x1; t[1] = y1; t[2] = x2; t[3] = y2; t[4] = tx1; t[5] = ty1; t[6] = bw; t[7] = bh;
}

// rerec(): square box around the centre
__device__ void square_box(double& x1, double& y1, double& x2, double& y2) {
    const double w = x2 - x1, h = y2 - y1, side = fmax(w, h);
    x1 = __dadd_rn(mad_sep(x1, w, 0.5), -__dmul_rn(side, 0.5));
    y1 = __dadd_rn(mad_sep(y1, h, 0.5), -__dmul_rn(side, 0.5));
    x2 = x1 + side;
    y2 = y1 + side;
}

// counters: [0] boxes found by stage 1 so far (all levels), [1] boxes after stage 1, [2] after stage 2, [3] after stage 3, [4] overflow
extern __shared__ unsigned char post_smem[];

// ---- stage 1, one pyramid level: threshold the face map, boxes of the firing cells, NMS 0.5, append the survivors ----------------
__global__ __launch_bounds__(NT) void stage1_level_kernel(const float* __restrict__ prob /*[w,h,2]*/, const float* __restrict__ reg /*[w,h,4]*/,
                                                          int w, int h, double scale, float thr, double* __restrict__ found /*[CAP,9]*/,
                                                          int* __restrict__ counters) {
    PostLds& L = *(PostLds*)post_smem;
    const int tid = threadIdx.x;
    if (tid == 0) L.n = 0;
    __syncthreads();
    const int cells = w * h;
    for (int i = tid; i < cells; i += NT) {
        const float s = prob[2 * i + 1];
        if (s >= thr) {
            const int pos = atomicAdd(&L.n, 1);
            if (pos < CAP) L.key[pos] = make_key(s, (unsigned)i);
        }
    }
    __syncthreads();
    const int n = L.n;
    if (n == 0) return;
    if (n > CAP) { if (tid == 0) counters[4] = 1; return; }
    const int p2 = np2_of(n);
    for (int i = n + tid; i < p2; i += NT) L.key[i] = ~0ull;
    __syncthreads();
    sort_keys(L, p2);
    for (int i = tid; i < n; i += NT) {
        const unsigned idx = key_idx(L.key[i]);
        const int xi = (int)(idx / (unsigned)h), yi = (int)(idx - (unsigned)xi * (unsigned)h);
        L.x1[i] = dfix((double)(2 * xi + 1) / scale);
        L.y1[i] = dfix((double)(2 * yi + 1) / scale);
        L.x2[i] = dfix((double)(2 * xi + 12) / scale);
        L.y2[i] = dfix((double)(2 * yi + 12) / scale);
        L.area[i] = __dmul_rn(L.x2[i] - L.x1[i] + 1.0, L.y2[i] - L.y1[i] + 1.0);
    }
    __syncthreads();
    nms_sorted(L, n, 0.5, false);
    const int base = counters[0], nk = L.nkeep;
    if (base + nk > CAP) { if (tid == 0) counters[4] = 1; return; }
    for (int k = tid; k < nk; k += NT) {
        const int i = L.keep_pos[k];
        const unsigned idx = key_idx(L.key[i]);
        const int xi = (int)(idx / (unsigned)h), yi = (int)(idx - (unsigned)xi * (unsigned)h);
        // the reference reads the regression maps flipped when exactly one cell fires (facial_analysis.py:383-387)
        const int rxi = n == 1 ? w - 1 - xi : xi;
        const float* r = reg + ((size_t)rxi * h + yi) * 4;
        double* o = found + (size_t)(base + k) * 9;
        o[0] = L.x1[i]; o[1] = L.y1[i]; o[2] = L.x2[i]; o[3] = L.y2[i];
        o[4] = (double)prob[2 * idx + 1];
        o[5] = (double)r[0]; o[6] = (double)r[1]; o[7] = (double)r[2]; o[8] = (double)r[3];
    }
    __syncthreads();
    if (tid == 0) counters[0] = base + nk;
}

// ---- stage 1, after the last level: NMS 0.7 over all survivors, regression by the P-Net offsets, square, truncate, crop windows ----
__global__ __launch_bounds__(NT) void stage1_finish_kernel(const double* __restrict__ found, int* __restrict__ counters, double* __restrict__ boxes /*[CAP,5]*/,
                                                           int* __restrict__ tab /*[CAP,8]*/, int img_w, int img_h) {
    PostLds& L = *(PostLds*)post_smem;
    const int tid = threadIdx.x;
    const int n = counters[0];
    if (n == 0 || counters[4]) { if (tid == 0) counters[1] = 0; return; }
    const int p2 = np2_of(n);
    for (int i = tid; i < p2; i += NT) L.key[i] = i < n ? make_key((float)found[(size_t)i * 9 + 4], (unsigned)i) : ~0ull;
    __syncthreads();
    sort_keys(L, p2);
    for (int i = tid; i < n; i += NT) {
        const double* b = found + (size_t)key_idx(L.key[i]) * 9;
        L.x1[i] = b[0]; L.y1[i] = b[1]; L.x2[i] = b[2]; L.y2[i] = b[3];
        L.area[i] = __dmul_rn(b[2] - b[0] + 1.0, b[3] - b[1] + 1.0);
    }
    __syncthreads();
    nms_sorted(L, n, 0.7, false);
    const int nk = L.nkeep;
    for (int k = tid; k < nk; k += NT) {
        const double* b = found + (size_t)key_idx(L.key[L.keep_pos[k]]) * 9;
        const double rw = b[2] - b[0], rh = b[3] - b[1];
        double x1 = mad_sep(b[0], b[5], rw), y1 = mad_sep(b[1], b[6], rh), x2 = mad_sep(b[2], b[7], rw), y2 = mad_sep(b[3], b[8], rh);
        square_box(x1, y1, x2, y2);
        double* o = boxes + (size_t)k * 5;
        o[0] = dfix(x1); o[1] = dfix(y1); o[2] = dfix(x2); o[3] = dfix(y2); o[4] = b[4];
        crop_row(o, img_w, img_h, tab + (size_t)k * 8);
    }
    if (tid == 0) counters[1] = nk;
}

// ---- stage 2 / 3 after the net: score filter, NMS 0.7, regression, (stage 2) square + truncate + crop windows, (stage 3) landmarks ----
// STAGE 2: boxes_out [m,5] = truncated square boxes (score column truncated too, as np.fix(boxes) does), tab_out rows.
// STAGE 3: regression BEFORE the NMS ('Min' overlap), boxes_out = regressed boxes + score, points_out [m,10] float32.
template <int STAGE>
__global__ __launch_bounds__(NT) void stage23_finish_kernel(const double* __restrict__ boxes_in /*[n,5]*/, int n, const float* __restrict__ prob /*[n,2]*/,
                                                            const float* __restrict__ reg /*[n,4]*/, const float* __restrict__ pts /*[n,10]*/, float thr,
                                                            double* __restrict__ boxes_out, int* __restrict__ tab_out, float* __restrict__ points_out,
                                                            int* __restrict__ counters, int img_w, int img_h) {
    PostLds& L = *(PostLds*)post_smem;
    const int tid = threadIdx.x;
    if (tid == 0) L.n = 0;
    __syncthreads();
    if (n > CAP) { if (tid == 0) { counters[4] = 1; counters[STAGE] = 0; } return; }
    for (int i = tid; i < n; i += NT) {
        const float s = prob[2 * i + 1];
        if (s > thr) L.key[atomicAdd(&L.n, 1)] = make_key(s, (unsigned)i);
    }
    __syncthreads();
    const int m = L.n;
    if (m == 0) { if (tid == 0) counters[STAGE] = 0; return; }
    const int p2 = np2_of(m);
    for (int i = m + tid; i < p2; i += NT) L.key[i] = ~0ull;
    __syncthreads();
    sort_keys(L, p2);
    for (int i = tid; i < m; i += NT) {
        const unsigned src = key_idx(L.key[i]);
        const double* b = boxes_in + (size_t)src * 5;
        double x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
        if (STAGE == 3) {      // bbreg before the NMS
            const double bw = x2 - x1 + 1.0, bh = y2 - y1 + 1.0;
            const float* r = reg + (size_t)src * 4;
            const double ox1 = x1, oy1 = y1, ox2 = x2, oy2 = y2;
            x1 = mad_sep(ox1, (double)r[0], bw); y1 = mad_sep(oy1, (double)r[1], bh);
            x2 = mad_sep(ox2, (double)r[2], bw); y2 = mad_sep(oy2, (double)r[3], bh);
        }
        L.x1[i] = x1; L.y1[i] = y1; L.x2[i] = x2; L.y2[i] = y2;
        L.area[i] = __dmul_rn(x2 - x1 + 1.0, y2 - y1 + 1.0);
    }
    __syncthreads();
    nms_sorted(L, m, 0.7, STAGE == 3);
    const int nk = L.nkeep;
    for (int k = tid; k < nk; k += NT) {
        const int i = L.keep_pos[k];
        const unsigned src = key_idx(L.key[i]);
        const float score = prob[2 * src + 1];
        double* o = boxes_out + (size_t)k * 5;
        if (STAGE == 2) {
            const double* b = boxes_in + (size_t)src * 5;
            const double bw = b[2] - b[0] + 1.0, bh = b[3] - b[1] + 1.0;
            const float* r = reg + (size_t)src * 4;
            double x1 = mad_sep(b[0], (double)r[0], bw), y1 = mad_sep(b[1], (double)r[1], bh);
            double x2 = mad_sep(b[2], (double)r[2], bw), y2 = mad_sep(b[3], (double)r[3], bh);
            square_box(x1, y1, x2, y2);
            o[0] = dfix(x1); o[1] = dfix(y1); o[2] = dfix(x2); o[3] = dfix(y2); o[4] = dfix((double)score);
            crop_row(o, img_w, img_h, tab_out + (size_t)k * 8);
        } else {
            o[0] = L.x1[i]; o[1] = L.y1[i]; o[2] = L.x2[i]; o[3] = L.y2[i]; o[4] = (double)score;
            const double* b = boxes_in + (size_t)src * 5;       // landmarks relative to the box BEFORE the regression
            const double bw = b[2] - b[0] + 1.0, bh = b[3] - b[1] + 1.0;
            const float* q = pts + (size_t)src * 10;
            float* po = points_out + (size_t)k * 10;
#pragma unroll
            for (int e = 0; e < 5; ++e) {
                po[e] = (float)__dadd_rn(__dadd_rn(__dmul_rn(bw, (double)q[e]), b[0]), -1.0);
                po[5 + e] = (float)__dadd_rn(__dadd_rn(__dmul_rn(bh, (double)q[5 + e]), b[1]), -1.0);
            }
        }
    }
    if (tid == 0) counters[STAGE] = nk;
}

// a plain NMS over a box list, for the unit tests: keep[] = kept ORIGINAL indices in pick order, *n_keep their number
__global__ __launch_bounds__(NT) void nms_kernel(const double* __restrict__ boxes /*[n,5]*/, int n, double thr, int use_min, int* __restrict__ keep,
                                                 int* __restrict__ n_keep) {
    PostLds& L = *(PostLds*)post_smem;
    const int tid = threadIdx.x;
    if (n == 0) { if (tid == 0) *n_keep = 0; return; }
    const int p2 = np2_of(n);
    for (int i = tid; i < p2; i += NT) L.key[i] = i < n ? make_key((float)boxes[(size_t)i * 5 + 4], (unsigned)i) : ~0ull;
    __syncthreads();
    sort_keys(L, p2);
    for (int i = tid; i < n; i += NT) {
        const double* b = boxes + (size_t)key_idx(L.key[i]) * 5;
        L.x1[i] = b[0]; L.y1[i] = b[1]; L.x2[i] = b[2]; L.y2[i] = b[3];
        L.area[i] = __dmul_rn(b[2] - b[0] + 1.0, b[3] - b[1] + 1.0);
    }
    __syncthreads();
    nms_sorted(L, n, thr, use_min != 0);
    for (int k = tid; k < L.nkeep; k += NT) keep[k] = (int)key_idx(L.key[L.keep_pos[k]]);
    if (tid == 0) *n_keep = L.nkeep;
}

template <class K>
int set_lds(K kernel) {      // (every launch: the attribute is per device, the calls are a few per frame)
    HSEFR_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PostLds)));
    return HSEFR_OK;
}

}  // namespace

int mtcnn_post_capacity() { return CAP; }

int launch_mtcnn_stage1_level(const float* prob, const float* reg, int w, int h, double scale, float thr, double* found, int* counters,
                              hipStream_t s) {
    HSEFR_REQUIRE(w > 0 && h > 0 && scale > 0 && (long long)w * h < (1ll << 31), HSEFR_ERR_INVALID, "mtcnn_stage1_level: bad map %dx%d", w, h);
    if (int rc = set_lds(stage1_level_kernel)) return rc;
    HSEFR_LAUNCH(stage1_level_kernel, dim3(1), dim3(NT), sizeof(PostLds), s, prob, reg, w, h, scale, thr, found, counters);
    return launch_status("mtcnn_stage1_level");
}

int launch_mtcnn_stage1_finish(const double* found, int* counters, double* boxes, int* tab, int img_w, int img_h, hipStream_t s) {
    if (int rc = set_lds(stage1_finish_kernel)) return rc;
    HSEFR_LAUNCH(stage1_finish_kernel, dim3(1), dim3(NT), sizeof(PostLds), s, found, counters, boxes, tab, img_w, img_h);
    return launch_status("mtcnn_stage1_finish");
}

int launch_mtcnn_stage23_finish(int stage, const double* boxes_in, int n, const float* prob, const float* reg, const float* pts, float thr,
                                double* boxes_out, int* tab_out, float* points_out, int* counters, int img_w, int img_h, hipStream_t s) {
    HSEFR_REQUIRE((stage == 2 || stage == 3) && n >= 0, HSEFR_ERR_INVALID, "mtcnn_stage_finish: stage %d, n %d", stage, n);
    if (stage == 2) {
        if (int rc = set_lds(stage23_finish_kernel<2>)) return rc;
        HSEFR_LAUNCH(stage23_finish_kernel<2>, dim3(1), dim3(NT), sizeof(PostLds), s, boxes_in, n, prob, reg, pts, thr, boxes_out, tab_out,
                           points_out, counters, img_w, img_h);
    } else {
        if (int rc = set_lds(stage23_finish_kernel<3>)) return rc;
        HSEFR_LAUNCH(stage23_finish_kernel<3>, dim3(1), dim3(NT), sizeof(PostLds), s, boxes_in, n, prob, reg, pts, thr, boxes_out, tab_out,
                           points_out, counters, img_w, img_h);
    }
    return launch_status("mtcnn_stage_finish");
}

int launch_mtcnn_nms(const double* boxes, int n, double thr, int use_min, int* keep, int* n_keep, hipStream_t s) {
    HSEFR_REQUIRE(n >= 0 && n <= CAP, HSEFR_ERR_UNSUPPORTED, "mtcnn_nms: %d boxes (capacity %d)", n, CAP);
    if (int rc = set_lds(nms_kernel)) return rc;
    HSEFR_LAUNCH(nms_kernel, dim3(1), dim3(NT), sizeof(PostLds), s, boxes, n, thr, use_min, keep, n_keep);
    return launch_status("mtcnn_nms");
}

}  // namespace hsefr
