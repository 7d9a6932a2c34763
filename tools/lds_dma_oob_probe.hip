// LDS-DMA out-of-bounds probe (DESIGN.md lesson 15a): what does `buffer_load_dwordx4 ... lds` write into LDS for a lane
// whose offset is outside the buffer resource?   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_oob_probe.hip -o /tmp/p && /tmp/p
// Result on MI355X: ZEROS (lanes 3, 8, 13, ... below, and the lane straddling the end), the in-range lanes their data with
// the SGPR offset applied.  csrc/dwpw_f16s.hip relies on it for the convolution's SAME padding.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, float* y, int n) {
    __shared__ __attribute__((aligned(1024))) float s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, n * 4, 0x00020000);
    unsigned voff = threadIdx.x * 16u;
    if (threadIdx.x % 5 == 3) voff = 0xFFFFFFF0u;          // "padding" lanes
    if (threadIdx.x == 60) voff = n * 4 - 8;                // straddles the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)s, 16, voff, 64 /*soffset*/, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) y[i] = s[i];
}
int main() {
    float *x, *y; const int n = 1024;
    hipMalloc(&x, n * 4); hipMalloc(&y, 256 * 4);
    float h[1024]; for (int i = 0; i < n; ++i) h[i] = i + 1;
    hipMemcpy(x, h, n * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(x, y, n);
    float o[256]; hipMemcpy(o, y, sizeof o, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %g %g %g %g%s", l, o[4*l], o[4*l+1], o[4*l+2], o[4*l+3], l % 2 ? "\n" : "   |   ");
    return 0;
}
