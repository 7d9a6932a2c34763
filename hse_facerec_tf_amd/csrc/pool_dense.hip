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

// Dense head: thread = output column, workgroup = DR rows of x staged in LDS; each weight is
// read once per DR rows (coalesced over columns), x values are LDS broadcasts.
constexpr int DR = 8;
__global__ __launch_bounds__(256) void dense_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ y, int n,
                                                    int k, int cout, int act) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [DR][k]
    const int r0 = blockIdx.x * DR;
    const int rows = min(DR, n - r0);
    for (int i = threadIdx.x; i < rows * k; i += 256) xs[i] = x[(size_t)r0 * k + i];
    for (int i = rows * k + threadIdx.x; i < DR * k; i += 256) xs[i] = 0.f;
    __syncthreads();
    const int col = blockIdx.y * 256 + threadIdx.x;
    if (col >= cout) return;
    float acc[DR];
#pragma unroll
    for (int r = 0; r < DR; ++r) acc[r] = 0.f;
    for (int kk = 0; kk < k; ++kk) {
        const float wv = w[(size_t)kk * cout + col];
#pragma unroll
        for (int r = 0; r < DR; ++r) acc[r] = fmaf(xs[r * k + kk], wv, acc[r]);
    }
    const float b = bias ? bias[col] : 0.f;
    for (int r = 0; r < rows; ++r) y[(size_t)(r0 + r) * cout + col] = apply_act_rt(acc[r] + b, act);
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

}  // namespace

int launch_gap(const float* x, float* y, int n, int hw, int c, hipStream_t s) {
    HSEFR_REQUIRE(c > 0 && c % 4 == 0, HSEFR_ERR_UNSUPPORTED, "gap: c=%d must be a multiple of 4", c);
    HSEFR_REQUIRE(n >= 0 && hw > 0, HSEFR_ERR_INVALID, "gap: bad shape");
    if (n == 0) return HSEFR_OK;
    const int c4 = c / 4;
    const long long waves = (long long)n * ((c4 + 15) / 16);
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipLaunchKernelGGL(gap_kernel, grid, block, 0, s, (const float4*)x, (float4*)y, n, hw, c4);
    return launch_status("gap");
}

int launch_dense(const float* x, const float* wgt, const float* bias, float* y, int n, int k, int cout, int act,
                 hipStream_t s) {
    HSEFR_REQUIRE(k > 0 && k <= 2048 && cout > 0, HSEFR_ERR_UNSUPPORTED, "dense: k=%d cout=%d", k, cout);
    HSEFR_REQUIRE(n >= 0, HSEFR_ERR_INVALID, "dense: n=%d", n);
    if (n == 0) return HSEFR_OK;
    dim3 grid((n + DR - 1) / DR, (cout + 255) / 256), block(256);
    const size_t lds = (size_t)DR * k * sizeof(float);
    hipLaunchKernelGGL(dense_kernel, grid, block, lds, s, x, wgt, bias, y, n, k, cout, act);
    return launch_status("dense");
}

int launch_softmax(const float* x, float* y, int n, int c, hipStream_t s) {
    HSEFR_REQUIRE(c > 0 && c <= 1024, HSEFR_ERR_UNSUPPORTED, "softmax: c=%d (max 1024)", c);
    HSEFR_REQUIRE(n >= 0, HSEFR_ERR_INVALID, "softmax: n=%d", n);
    if (n == 0) return HSEFR_OK;
    dim3 grid((n + 3) / 4), block(256);
    hipLaunchKernelGGL(softmax_kernel, grid, block, 0, s, x, y, n, c);
    return launch_status("softmax");
}

}  // namespace hsefr
