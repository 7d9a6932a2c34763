// Probe: how fast can ONE CU issue 16-byte-per-lane stores?  (tools/; build: hipcc --offload-arch=gfx950 -O3 tools/store_rate_probe.hip -o /tmp/srp)
// Each workgroup owns a 144 KB output region (a 288 x 128 fp32 tile, row pitch `pitch` bytes) and `waves` of its waves store
// it `reps` times; patterns: 0 = whole 128-B lines (8 lanes per row), 1 = half lines (4 lanes per row, 16 rows per
// instruction), 2 = whole lines but rows 2 KB apart inside one instruction (a GEMM epilogue's rows), 3 = dword stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(768, 1) void probe(float* out, unsigned long long* cyc, int waves, int reps, int pattern, unsigned pitch_bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = (char*)out + (size_t)blockIdx.x * 288 * pitch_bytes;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < waves) {
        f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
        for (int r = 0; r < reps; ++r) {
            // the wave's share: rows [wave * 288 / waves ...): 36 KB / (waves/8) ... generic: instruction i covers 1 KB
            const int n_inst = 144 / waves;          // 1-KB instructions per wave per repetition
            for (int i = 0; i < n_inst; ++i) {
                const int k = wave * n_inst + i;     // 0 .. 143: which KB of the tile
                char* p;
                if (pattern == 0) p = base + (size_t)(k * 2 + (lane >> 5)) * pitch_bytes + (k % 4) * 0 + (lane & 31) * 16 % 512;   // 2 rows x 512 B
                else if (pattern == 1) p = base + (size_t)((k >> 3) * 16 + (lane & 15)) * pitch_bytes + (k & 7) * 64 + (lane >> 4) * 16;   // 16 rows x 64 B
                else if (pattern == 2) p = base + (size_t)((k >> 2) * 8 + (lane >> 3)) * pitch_bytes + (k & 3) * 128 + (lane & 7) * 16;   // 8 rows x 128 B
                else p = base + (size_t)(k * 2 + (lane >> 5)) * pitch_bytes + (lane & 31) * 16;
                if (pattern == 3) {
                    float* q = (float*)(base + (size_t)k * 1024);
                    q[lane] = v.x; q[64 + lane] = v.y; q[128 + lane] = v.z; q[192 + lane] = v.w;
                } else {
                    *(f32x4*)p = v;
                }
                v.x += 1.f;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (lane == 0 && wave < waves) cyc[gridDim.x + blockIdx.x * 12 + wave] = t1 - t0;
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    float* out; unsigned long long* cyc;
    const unsigned pitch = 2048;
    hipMalloc(&out, (size_t)grid * 288 * pitch);
    hipMalloc(&cyc, (grid * 13) * 8);
    unsigned long long* h = (unsigned long long*)malloc(grid * 13 * 8);
    const int reps = 8;
    for (int pattern = 0; pattern < 4; ++pattern)
        for (int waves : {1, 2, 4, 8, 12}) {
            if (144 % waves) continue;
            for (int it = 0; it < 2; ++it) {
                hipLaunchKernelGGL(probe, dim3(grid), dim3(768), 0, 0, out, cyc, waves, reps, pattern, pitch);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, grid * 13 * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (int b = 0; b < grid; ++b) s += (double)h[grid + b * 12];
            s /= grid;
            printf("grid %3d pattern %d waves %2d: %8.0f cycles per wave for %d x %d KB -> %6.1f cycles per 1-KB store, %5.1f B/clk/CU\n", grid, pattern, waves,
                   s, reps, 144 / waves, s / (reps * (144 / waves)), 144.0 * 1024 * reps / s);
        }
    return 0;
}
