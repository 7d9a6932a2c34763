// Probe: is a raw buffer_load_dwordx4 that straddles the END of its resource range-checked per DWORD (in-range dwords
// returned, the rest zero) or dropped as a whole?  And one that starts BEFORE offset 0 (negative = wrapped offset)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* x, float* out, unsigned bytes) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, bytes, 0x00020000);
    const unsigned offs[4] = {bytes - 8u, bytes - 4u, bytes, (unsigned)-8};
    for (int i = 0; i < 4; ++i) {
        f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, offs[i], 0, 0));
        for (int e = 0; e < 4; ++e) out[i * 4 + e] = v[e];
    }
}
int main() {
    float h[64], *d, *o, ho[16];
    for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
    hipMalloc(&d, 256); hipMalloc(&o, 64);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    const unsigned bytes = 27 * 4;      // resource covers floats 0..26 (not a multiple of 16 bytes)
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, d, o, bytes);
    hipMemcpy(ho, o, 64, hipMemcpyDeviceToHost);
    const char* what[4] = {"starts 8 B before the end ", "starts 4 B before the end ", "starts at the end         ", "starts 8 B before offset 0"};
    for (int i = 0; i < 4; ++i) printf("%s: %6.1f %6.1f %6.1f %6.1f\n", what[i], ho[4 * i], ho[4 * i + 1], ho[4 * i + 2], ho[4 * i + 3]);
    printf("(floats 25, 26 = 125, 126 are the last two inside the resource; 127.. are outside)\n");
    return 0;
}
