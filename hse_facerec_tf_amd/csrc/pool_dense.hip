// Global average pool, dense heads, softmax -- the tail of the graph (nodes #230-241), gfx950.
//   global_pooling/Mean (#230)                        -> gap_kernel
//   feats/MatMul + BiasAdd + Relu (#232-234)          -> dense_kernel (ACT_RELU)
//   gender_pred/MatMul + BiasAdd -> Sigmoid (#236-238) -> dense_kernel (ACT_SIGMOID)
//   age_pred/MatMul + BiasAdd (#239-240)              -> dense_kernel (ACT_NONE)
//   age_pred/Softmax (#241)                           -> softmax_kernel
// run by sess.run at facial_analysis.py:109 (facerec_test.py:120 fetches only the GAP).
#include "common.h"

namespace hsefr {

namespace {

__device__ __forceinline__ float4 add4(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 shfl_xor4(float4 v, int m) {
    return make_float4(__shfl_xor(v.x, m), __shfl_xor(v.y, m), __shfl_xor(v.z, m), __shfl_xor(v.w, m));
}

// One wave = 64 channels (16 float4 lanes) x 4 interleaved slices of the H*W positions; the
// four partial sums meet through two wavefront shuffles (xor 16, xor 32).  Every load is
// 16 B/lane with 256-B contiguous runs; HBM-bound (reads hw*c*4 B, writes c*4 B per image).
__global__ __launch_bounds__(256) void gap_kernel(const float4* __restrict__ x, float4* __restrict__ y, int n,
                                                  int hw, int c4) {
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int groups = (c4 + 15) / 16;  // 64-channel groups per image
    if (wave_global >= n * groups) return;
    const int img = wave_global / groups;
    const int grp = wave_global - img * groups;
    const int cq = grp * 16 + (lane & 15);
    const int part = lane >> 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cq < c4) {
        const float4* p = x + (size_t)img * hw * c4 + cq;
        for (int i = part; i < hw; i += 4) acc = add4(acc, p[(size_t)i * c4]);
    }
    acc = add4(acc, shfl_xor4(acc, 16));
    acc = add4(acc, shfl_xor4(acc, 32));
    if (part == 0 && cq < c4) {
        const float d = (float)hw;
        y[(size_t)img * c4 + cq] = make_float4(acc.x / d, acc.y / d, acc.z / d, acc.w / d);
    }
}

// Dense head: a workgroup = DR rows of x (staged in LDS) x 64 output columns, the contraction split over its four waves (each
// takes a quarter of k; the partial sums meet in LDS and are added in wave order: deterministic).  Each weight is read once per
// DR rows, coalesced over the 64 columns of a wave; x values are LDS broadcasts.  (Round 3: the first version gave a workgroup 256
// columns and the whole of k -- 64 workgroups and 1024 serial steps per thread for the 512 x 1024 x 256 head of BASELINE config 4:
// 122 us, 3 % of that step; this form fills the chip at every batch size and runs it in ~10 us.)
constexpr int DR = 8, DCOLS = 64;
__global__ __launch_bounds__(256) void dense_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ y, int n,
                                                    int k, int cout, int act) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [DR][k], then the partial sums [4][DR][DCOLS]
    const int r0 = blockIdx.x * DR;
    const int rows = min(DR, n - r0);
    if ((k & 3) == 0) {                 // 16-byte loads, eight in flight per thread (a load -> LDS-write loop pays one latency per trip)
        const float4* xg = (const float4*)(x + (size_t)r0 * k);
        const int nq = rows * k / 4, tq = DR * k / 4;
        for (int i0 = 0; i0 < tq; i0 += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256 + threadIdx.x;
                v[u] = i < nq ? xg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256 + threadIdx.x;
                if (i < tq) ((float4*)xs)[i] = v[u];
            }
        }
    } else {
        for (int i = threadIdx.x; i < rows * k; i += 256) xs[i] = x[(size_t)r0 * k + i];
        for (int i = rows * k + threadIdx.x; i < DR * k; i += 256) xs[i] = 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.y * DCOLS + lane;
    const int kq = (k + 3) >> 2, k0 = wave * kq, k1 = min(k, k0 + kq);
    float acc[DR];
#pragma unroll
    for (int r = 0; r < DR; ++r) acc[r] = 0.f;
    if (col < cout) {
        int kk = k0;
        if ((k & 3) == 0 && (kq & 3) == 0) {          // 16 steps at a time: sixteen weights in flight, one 16-byte LDS broadcast per row and 4 steps
            for (; kk + 16 <= k1; kk += 16) {
                float wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = w[(size_t)(kk + u) * cout + col];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < DR; ++r) {
                        const float4 xv = *(const float4*)(xs + r * k + kk + 4 * g);
                        acc[r] = fmaf(xv.w, wv[4 * g + 3], fmaf(xv.z, wv[4 * g + 2], fmaf(xv.y, wv[4 * g + 1], fmaf(xv.x, wv[4 * g], acc[r]))));
                    }
            }
            for (; kk + 4 <= k1; kk += 4) {
                const float w0 = w[(size_t)kk * cout + col], w1 = w[(size_t)(kk + 1) * cout + col];
                const float w2 = w[(size_t)(kk + 2) * cout + col], w3 = w[(size_t)(kk + 3) * cout + col];
#pragma unroll
                for (int r = 0; r < DR; ++r) {
                    const float4 xv = *(const float4*)(xs + r * k + kk);
                    acc[r] = fmaf(xv.w, w3, fmaf(xv.z, w2, fmaf(xv.y, w1, fmaf(xv.x, w0, acc[r]))));
                }
            }
        }
        for (; kk < k1; ++kk) {
            const float wv = w[(size_t)kk * cout + col];
#pragma unroll
            for (int r = 0; r < DR; ++r) acc[r] = fmaf(xs[r * k + kk], wv, acc[r]);
        }
    }
    float* part = xs + DR * k;
#pragma unroll
    for (int r = 0; r < DR; ++r) part[(wave * DR + r) * DCOLS + lane] = acc[r];
    __syncthreads();
    for (int o = threadIdx.x; o < DR * DCOLS; o += 256) {
        const int r = o / DCOLS, c = o - r * DCOLS, oc = blockIdx.y * DCOLS + c;
        if (r < rows && oc < cout) {
            const float sum = ((part[(0 * DR + r) * DCOLS + c] + part[(1 * DR + r) * DCOLS + c]) + part[(2 * DR + r) * DCOLS + c]) +
                              part[(3 * DR + r) * DCOLS + c];
            y[(size_t)(r0 + r) * cout + oc] = apply_act_rt(sum + (bias ? bias[oc] : 0.f), act);
        }
    }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// One wave per row, max-subtracted like tf.nn.softmax.
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int c) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* xr = x + (size_t)row * c;
    float v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = lane + 64 * i;
        v[i] = j < c ? xr[j] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = (lane + 64 * i) < c ? expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = lane + 64 * i;
        if (j < c) y[(size_t)row * c + j] = v[i] / sum;
    }
}


// ---- the age / gender heads in one launch (round 6) -------------------------------------------------------------------------------
// feats (k -> 256, ReLU), age_pred (256 -> a) + softmax, gender_pred (256 -> 1) + sigmoid: four launches of 0.58 MFLOP per image -- 63 us
// of the 3.2 ms batch-512 step (profiles/r05_agegender_layers.txt).  What they cost is the way dense_kernel reads its weights: a dword
// per lane and k (256 bytes per wave instruction), 1 MB of w1 per 8 rows, 17 B/clk of the CU's vector-memory path.  (Measured first: the
// same reads with all four heads in one workgroup and 16 or 64 weights in flight per lane -- 33 us either way, 49 with 8 rows per
// workgroup: not latency, instruction count.)  Here a lane owns FOUR ADJACENT COLUMNS -- one 16-byte load per k, a wave instruction moves
// a whole 1 KiB row of w1 -- and the sixteen waves of a workgroup split k sixteen ways for all 256 columns; the partial sums meet in LDS
// and are added in slice order (fixed: bit-identical run to run and independent of the batch; another summation order than
// dense_kernel's four slices, same fp32 grade).  hidden and logits are written too (per-layer tests read them).
constexpr int HR = 4, HID = 256, HW = 16;        // rows per workgroup, hidden width, waves = slices of k
__global__ __launch_bounds__(1024) void heads_kernel(const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
                                                     const float* __restrict__ wa, const float* __restrict__ ba, const float* __restrict__ wg,
                                                     const float* __restrict__ bg, float* __restrict__ hidden, float* __restrict__ logits,
                                                     float* __restrict__ probs, float* __restrict__ gender, int n, int k, int a) {
    extern __shared__ __attribute__((aligned(16))) float hs[];   // xs [HR][k] | part [HW][HR][HID] | hid [HR][HID] | lg [HR][128] | pg [HW][HR]
    float* xs = hs;
    float* part = xs + HR * k;
    float* hid = part + HW * HR * HID;
    float* lg = hid + HR * HID;
    float* pg = lg + HR * 128;
    const int r0 = blockIdx.x * HR;
    const int rows = min(HR, n - r0);
    {
        const float4* xg = (const float4*)(x + (size_t)r0 * k);
        const int nq = rows * k / 4, tq = HR * k / 4;
        for (int i = threadIdx.x; i < tq; i += 1024) ((float4*)xs)[i] = i < nq ? xg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {   // hidden: slice `wave` of k, columns 4 lane .. 4 lane + 3
        const int ks = k / HW, k0 = wave * ks;
        float4 acc[HR];
#pragma unroll
        for (int r = 0; r < HR; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* wq = (const float4*)w1 + lane;              // row kk: wq[kk * 64]
        for (int kk = k0; kk < k0 + ks; kk += 8) {              // eight 1-KiB rows of w1 in flight per wave, sixteen waves per CU
            float4 wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = wq[(size_t)(kk + u) * 64];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < HR; ++r) {
                    const float4 xv = *(const float4*)(xs + r * k + kk + 4 * g);
                    const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float4 w = wv[4 * g + e];
                        acc[r].x = fmaf(xe[e], w.x, acc[r].x);
                        acc[r].y = fmaf(xe[e], w.y, acc[r].y);
                        acc[r].z = fmaf(xe[e], w.z, acc[r].z);
                        acc[r].w = fmaf(xe[e], w.w, acc[r].w);
                    }
                }
        }
#pragma unroll
        for (int r = 0; r < HR; ++r) *(float4*)(part + (wave * HR + r) * HID + 4 * lane) = acc[r];
    }
    __syncthreads();
    {
        const int r = threadIdx.x >> 8, c = threadIdx.x & 255;     // HR * HID = 1024 values, one per thread
        float sum = part[r * HID + c];
#pragma unroll
        for (int w = 1; w < HW; ++w) sum += part[(w * HR + r) * HID + c];
        const float v = fmaxf(sum + b1[c], 0.f);
        hid[r * HID + c] = v;
        if (r < rows) hidden[(size_t)(r0 + r) * HID + c] = v;
    }
    __syncthreads();
    {   // age logits (columns lane, lane + 64) and the gender logit: slice `wave` of the 256 hidden values (16 each)
        const int k0 = wave * (HID / HW);
        float acc0[HR], acc1[HR], accg[HR];
#pragma unroll
        for (int r = 0; r < HR; ++r) acc0[r] = acc1[r] = accg[r] = 0.f;
        const int c0 = lane, c1 = lane + 64;
#pragma unroll
        for (int u = 0; u < HID / HW; ++u) {
            const float w0 = c0 < a ? wa[(size_t)(k0 + u) * a + c0] : 0.f, w1v = c1 < a ? wa[(size_t)(k0 + u) * a + c1] : 0.f;
            const float wgv = wg[k0 + u];
#pragma unroll
            for (int r = 0; r < HR; ++r) {
                const float h = hid[r * HID + k0 + u];
                acc0[r] = fmaf(h, w0, acc0[r]);
                acc1[r] = fmaf(h, w1v, acc1[r]);
                accg[r] = fmaf(h, wgv, accg[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            part[(wave * HR + r) * 128 + c0] = acc0[r];
            part[(wave * HR + r) * 128 + c1] = acc1[r];
            if (lane == 0) pg[wave * HR + r] = accg[r];
        }
    }
    __syncthreads();
    if (threadIdx.x < HR * 128) {
        const int r = threadIdx.x >> 7, c = threadIdx.x & 127;
        if (c < a) {
            float sum = part[r * 128 + c];
#pragma unroll
            for (int w = 1; w < HW; ++w) sum += part[(w * HR + r) * 128 + c];
            const float v = sum + ba[c];
            lg[r * 128 + c] = v;
            if (r < rows) logits[(size_t)(r0 + r) * a + c] = v;
        }
    } else if (threadIdx.x < HR * 128 + HR) {
        const int r = threadIdx.x - HR * 128;
        float sum = pg[r];
#pragma unroll
        for (int w = 1; w < HW; ++w) sum += pg[w * HR + r];
        if (r < rows) gender[r0 + r] = apply_act_rt(sum + bg[0], HSEFR_ACT_SIGMOID);
    }
    __syncthreads();
    if (wave < rows) {       // softmax_kernel's row code on the logits in LDS (a <= 128: two values per lane)
        const float* xr = lg + wave * 128;
        float v[2];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = lane + 64 * i;
            v[i] = j < a ? xr[j] : -INFINITY;
            mx = fmaxf(mx, v[i]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v[i] = (lane + 64 * i) < a ? expf(v[i] - mx) : 0.f;
            sum += v[i];
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = lane + 64 * i;
            if (j < a) probs[(size_t)(r0 + wave) * a + j] = v[i] / sum;
        }
    }
}

}  // namespace

int launch_gap(const float* x, float* y, int n, int hw, int c, hipStream_t s) {
    HSEFR_REQUIRE(c > 0 && c % 4 == 0, HSEFR_ERR_UNSUPPORTED, "gap: c=%d must be a multiple of 4", c);
    HSEFR_REQUIRE(n >= 0 && hw > 0, HSEFR_ERR_INVALID, "gap: bad shape");
    if (n == 0) return HSEFR_OK;
    const int c4 = c / 4;
    const long long waves = (long long)n * ((c4 + 15) / 16);
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    HSEFR_LAUNCH(gap_kernel, grid, block, 0, s, (const float4*)x, (float4*)y, n, hw, c4);
    return launch_status("gap");
}

int launch_dense(const float* x, const float* wgt, const float* bias, float* y, int n, int k, int cout, int act,
                 hipStream_t s) {
    HSEFR_REQUIRE(k > 0 && k <= 2048 && cout > 0, HSEFR_ERR_UNSUPPORTED, "dense: k=%d cout=%d", k, cout);
    HSEFR_REQUIRE(n >= 0, HSEFR_ERR_INVALID, "dense: n=%d", n);
    if (n == 0) return HSEFR_OK;
    dim3 grid((n + DR - 1) / DR, (cout + DCOLS - 1) / DCOLS), block(256);
    const size_t lds = ((size_t)DR * k + 4 * DR * DCOLS) * sizeof(float);
    HSEFR_LAUNCH(dense_kernel, grid, block, lds, s, x, wgt, bias, y, n, k, cout, act);
    return launch_status("dense");
}

int launch_softmax(const float* x, float* y, int n, int c, hipStream_t s) {
    HSEFR_REQUIRE(c > 0 && c <= 1024, HSEFR_ERR_UNSUPPORTED, "softmax: c=%d (max 1024)", c);
    HSEFR_REQUIRE(n >= 0, HSEFR_ERR_INVALID, "softmax: n=%d", n);
    if (n == 0) return HSEFR_OK;
    dim3 grid((n + 3) / 4), block(256);
    HSEFR_LAUNCH(softmax_kernel, grid, block, 0, s, x, y, n, c);
    return launch_status("softmax");
}

int launch_heads_fused(const float* x, const float* w1, const float* b1, const float* wa, const float* ba, const float* wg, const float* bg,
                       float* hidden, float* logits, float* age_probs, float* gender, int n, int k, int a, hipStream_t s) {
    HSEFR_REQUIRE(k > 0 && k % 256 == 0 && k <= 2048 && a >= 1 && a <= 128, HSEFR_ERR_UNSUPPORTED, "heads_fused: k=%d a=%d (k %% 256 == 0, k <= 2048, a <= 128)", k, a);
    HSEFR_REQUIRE(n >= 0, HSEFR_ERR_INVALID, "heads_fused: n=%d", n);
    if (n == 0) return HSEFR_OK;
    const size_t lds = ((size_t)HR * k + HW * HR * HID + HR * HID + HR * 128 + HW * HR) * sizeof(float);
    HSEFR_LAUNCH(heads_kernel, dim3((n + HR - 1) / HR), dim3(1024), lds, s, x, w1, b1, wa, ba, wg, bg, hidden, logits, age_probs, gender, n, k, a);
    return launch_status("heads_fused");
}

}  // namespace hsefr
