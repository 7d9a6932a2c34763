#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, int n, int soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n * 4, 0x00020000);
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, soff, 0);
    f32x4 f = __builtin_bit_cast(f32x4, v);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    ((f32x4*)y)[threadIdx.x] = f * 2.0f;
}
