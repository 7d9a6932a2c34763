// Calibration kernels (tuning/debug only): what a plain float4 copy reaches on this GPU, i.e. the
// practical HBM ceiling the streaming kernels are compared against (MI355X_MICROARCH.md quotes
// 6.29 TB/s for a float4 copy against the 8.0 TB/s spec).
#include "common.h"

namespace hsefr {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// variant bits: 1 = nontemporal loads, 2 = nontemporal stores; U = float4s in flight per thread
template <int U, int NT>
__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256;
            if (i < n) v[u] = (NT & 1) ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256;
            if (i < n) {
                if (NT & 2) __builtin_nontemporal_store(v[u], dst + i);
                else dst[i] = v[u];
            }
        }
    }
}

// Clock probe (MI355X_MICROARCH.md "DVFS give-back" item 6): a dense fp32-MFMA loop on every CU, stamped with
// s_memtime (shader clock) and s_memrealtime (100 MHz): clock = d(memtime)/d(memrealtime) * 100 MHz.  The stamps go
// to a buffer nothing else reads.  mode 0: one wave per SIMD, 4 independent accumulators (peak issue rate);
// the achieved MFMA rate is also derivable: iters * 4 MFMAs * 4096 FLOP per wave.
typedef float f32x16p __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* __restrict__ out, int iters, float seed) {
    f32x16p a0, a1, a2, a3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = seed * r; a1[r] = seed + r; a2[r] = seed - r; a3[r] = seed * 2 + r; }
    const float x = seed + (threadIdx.x & 63) * 0.001f, y = seed - (threadIdx.x & 31) * 0.002f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sink += a0[r] + a1[r] + a2[r] + a3[r];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3 + 0] = t1 - t0;
        out[blockIdx.x * 3 + 1] = r1 - r0;
        out[blockIdx.x * 3 + 2] = (unsigned long long)__float_as_uint(sink);
    }
}

// Same probe with the f16 MFMA (v_mfma_f32_32x32x16_f16) on pseudo-random operands: the clock and matrix rate the
// split-f16 GEMMs can hope for (a dense 16-bit MFMA stream makes the chip lower its clock far more than the fp32 one).
typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void clock_probe_f16_kernel(unsigned long long* __restrict__ out, int iters, float seed) {
    f32x16p a0, a1, a2, a3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = seed * r; a1[r] = seed + r; a2[r] = seed - r; a3[r] = seed * 2 + r; }
    f16x8p x, y;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h = h * 1664525u + 1013904223u;
        x[j] = (_Float16)(((int)(h >> 9) & 0xffff) * (1.f / 65536.f) - 0.5f);
        h = h * 1664525u + 1013904223u;
        y[j] = (_Float16)(((int)(h >> 9) & 0xffff) * (1.f / 65536.f) - 0.5f);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, y, a3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sink += a0[r] + a1[r] + a2[r] + a3[r];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3 + 0] = t1 - t0;
        out[blockIdx.x * 3 + 1] = r1 - r0;
        out[blockIdx.x * 3 + 2] = (unsigned long long)__float_as_uint(sink);
    }
}

// ... and with the 16x16x32 shape (same FLOPs per cycle on paper; the chip holds a different clock under it)
typedef float f32x4p __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void clock_probe_f16s_kernel(unsigned long long* __restrict__ out, int iters, float seed) {
    f32x4p a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (f32x4p){seed * i, seed + i, seed - i, seed * 2 + i};
    f16x8p x, y;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h = h * 1664525u + 1013904223u;
        x[j] = (_Float16)(((int)(h >> 9) & 0xffff) * (1.f / 65536.f) - 0.5f);
        h = h * 1664525u + 1013904223u;
        y[j] = (_Float16)(((int)(h >> 9) & 0xffff) * (1.f / 65536.f) - 0.5f);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16((k & 1) ? x : y, (k & 2) ? x : y, a[k], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sink += a[k][0] + a[k][1] + a[k][2] + a[k][3];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3 + 0] = t1 - t0;
        out[blockIdx.x * 3 + 1] = r1 - r0;
        out[blockIdx.x * 3 + 2] = (unsigned long long)__float_as_uint(sink);
    }
}

int g_clock_mode = 0;    // hsefr_debug_set "clock_mode": 0 = fp32 MFMA probe, 1 = f16 32x32x16 probe, 2 = f16 16x16x32 probe
int g_copy_variant = 0;  // unroll: (v & 3) -> {1, 2, 4, 8}; nt bits: (v >> 2) & 3; grid: (v >> 4) & 3 -> {8, 4, 16, 32} WG/CU

template <int U, int NT>
void launch_v(const f32x4* s, f32x4* d, size_t n, unsigned blocks, hipStream_t st) {
    HSEFR_LAUNCH((copy_kernel<U, NT>), dim3(blocks), dim3(256), 0, st, s, d, n);
}

template <int U>
void launch_u(int nt, const f32x4* s, f32x4* d, size_t n, unsigned blocks, hipStream_t st) {
    switch (nt) {
        case 0: launch_v<U, 0>(s, d, n, blocks, st); break;
        case 1: launch_v<U, 1>(s, d, n, blocks, st); break;
        case 2: launch_v<U, 2>(s, d, n, blocks, st); break;
        default: launch_v<U, 3>(s, d, n, blocks, st); break;
    }
}

}  // namespace

void set_copy_variant(int v) { g_copy_variant = v; }
void set_clock_mode(int v) { g_clock_mode = v; }

int launch_clock_probe(unsigned long long* out, int blocks, int iters, hipStream_t s) {
    HSEFR_REQUIRE(out && blocks > 0 && iters > 0, HSEFR_ERR_INVALID, "clock_probe: bad argument");
    if (g_clock_mode == 1) HSEFR_LAUNCH(clock_probe_f16_kernel, dim3(blocks), dim3(256), 0, s, out, iters, 0.5f);
    else if (g_clock_mode == 2) HSEFR_LAUNCH(clock_probe_f16s_kernel, dim3(blocks), dim3(256), 0, s, out, iters, 0.5f);
    else HSEFR_LAUNCH(clock_probe_kernel, dim3(blocks), dim3(256), 0, s, out, iters, 0.5f);
    return launch_status("clock_probe");
}

int launch_copy(const void* src, void* dst, size_t bytes, hipStream_t s) {
    HSEFR_REQUIRE(bytes % 16 == 0, HSEFR_ERR_INVALID, "copy: bytes must be a multiple of 16");
    if (!bytes) return HSEFR_OK;
    const size_t n = bytes / 16;
    const int v = g_copy_variant;
    const int u = 1 << (v & 3), nt = (v >> 2) & 3;
    const unsigned per_cu[4] = {8, 4, 16, 32};
    size_t blocks = (n + (size_t)256 * u - 1) / ((size_t)256 * u);
    const size_t cap = 256u * per_cu[(v >> 4) & 3];
    if (blocks > cap) blocks = cap;
    const f32x4* sp = (const f32x4*)src;
    f32x4* dp = (f32x4*)dst;
    switch (u) {
        case 1: launch_u<1>(nt, sp, dp, n, (unsigned)blocks, s); break;
        case 2: launch_u<2>(nt, sp, dp, n, (unsigned)blocks, s); break;
        case 4: launch_u<4>(nt, sp, dp, n, (unsigned)blocks, s); break;
        default: launch_u<8>(nt, sp, dp, n, (unsigned)blocks, s); break;
    }
    return launch_status("copy");
}

}  // namespace hsefr
