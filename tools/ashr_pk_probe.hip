// Probe: what does v_ashr_pk_u8_i32 (new on gfx950) leave in the UPPER half of its destination register?
// hipcc (ROCm 7.2) pattern-matches  sat_u8(a >> n) | sat_u8(b >> n) << 8  into this instruction and then ORs further bytes into
// bits 16..31 of the result as if they were zero.  Build: hipcc --offload-arch=gfx950 -O2 -o ashr_pk_probe.bin ashr_pk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned* out, int a, int b, unsigned preset) {
    unsigned d = preset;
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 22" : "+v"(d) : "v"(a), "v"(b));
    out[0] = d;
    // what the compiler emits for the C expression (checked separately: an opaque zero keeps the pattern alive)
    int x = a + (int)out[1], y = b + (int)out[1], z = a + (int)out[2];
    const unsigned lo = (unsigned)min(max(x >> 22, 0), 255) | ((unsigned)min(max(y >> 22, 0), 255) << 8);
    out[3] = lo | ((unsigned)min(max(z >> 22, 0), 255) << 16);
}

int main() {
    unsigned* d;
    hipMalloc(&d, 16);
    hipMemset(d, 0, 16);
    const int a = 77 << 22, b = 200 << 22;
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, d, a, b, 0xDEADBEEFu);
    unsigned h[4];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("v_ashr_pk_u8_i32 d, a=77<<22, b=200<<22, 22 with d preset to 0xDEADBEEF -> 0x%08X\n", h[0]);
    printf("  low half  : 0x%04X (expected 0xC84D = {sat_u8(b>>22), sat_u8(a>>22)})\n", h[0] & 0xffff);
    printf("  upper half: 0x%04X (%s)\n", h[0] >> 16, (h[0] >> 16) == 0xDEAD ? "PRESERVED: the instruction writes 16 bits only" : ((h[0] >> 16) == 0 ? "zeroed" : "something else"));
    printf("C expression sat(a>>22) | sat(b>>22)<<8 | sat(a>>22)<<16 -> 0x%08X (expected 0x004DC84D)\n", h[3]);
    return 0;
}
