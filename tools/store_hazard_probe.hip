// gfx950 store-data hazard probe (DESIGN.md lesson 14; tools/isa_lint.py is the build-time guard that follows from it).
//
//   hipcc --offload-arch=gfx950 -O2 tools/store_hazard_probe.hip -o /tmp/probe && /tmp/probe
//
// Question: if the instruction right behind a `buffer_store_dwordx4` writes one of the store's data VGPRs, which value
// reaches memory?  LLVM's hazard recogniser (GCNHazardRecognizer::createsVALUHazard) gives such a store one wait state
// only when its soffset is an immediate, assuming the SGPR-soffset form is safe.  Each thread stores four 16-byte
// chunks of a known value; the instruction behind every store overwrites part of its data.  Any dword in memory that
// differs from the known value was replaced after the store had issued.
//
// Result on MI355X (profiles/r01_store_hazard_probe.txt): with an SGPR soffset a packed-math write (v_pk_*) behind the
// store replaces the LAST dwords in ~24 % of the stores, a VOP3 write the FIRST dword; one `s_nop 0` in between -> 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define FILL_A "v_mov_b32 v4, %1\n v_mov_b32 v5, %1\n v_mov_b32 v6, %1\n v_mov_b32 v7, %1\n s_nop 7\n"
#define FILL_B "v_mov_b32 v8, %1\n v_mov_b32 v9, %1\n v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n s_nop 7\n"
// four stores; behind them: v_pk_mul on dwords 2-3 | v_mov on dword 3 | v_bfe (VOP3) on dword 0 | v_pk_mul on dwords 2-3 from other sources
#define BODY(SO0, SO1, SO2, SO3, GAP)                                                                                  \
    FILL_A FILL_B                                                                                                     \
    "buffer_store_dwordx4 v[4:7], %0, %2, " SO0 "\n" GAP "v_pk_mul_f32 v[6:7], v[6:7], v[6:7]\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n" \
    "buffer_store_dwordx4 v[8:11], %0, %2, " SO1 "\n" GAP "v_mov_b32 v11, 0\n v_mov_b32 v10, 0\n v_mov_b32 v9, 0\n v_mov_b32 v8, 0\n"     \
    FILL_A                                                                                                            \
    "buffer_store_dwordx4 v[4:7], %0, %2, " SO2 "\n" GAP "v_bfe_u32 v4, v5, 16, 1\n v_mov_b32 v7, 0\n v_mov_b32 v6, 0\n v_mov_b32 v5, 0\n" \
    FILL_B                                                                                                            \
    "buffer_store_dwordx4 v[8:11], %0, %2, " SO3 "\n" GAP "v_pk_mul_f32 v[10:11], v[8:9], v[8:9]\n v_mov_b32 v8, 0\n"    \
    "s_waitcnt vmcnt(0)\n"

// global_store form of the same sequence (%7 = 64-bit address pair)
#define GBODY(GAP)                                                                                                     \
    FILL_A FILL_B                                                                                                     \
    "global_store_dwordx4 %7, v[4:7], off\n" GAP "v_pk_mul_f32 v[6:7], v[6:7], v[6:7]\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n"          \
    "global_store_dwordx4 %7, v[8:11], off offset:16\n" GAP "v_mov_b32 v11, 0\n v_mov_b32 v10, 0\n v_mov_b32 v9, 0\n v_mov_b32 v8, 0\n" \
    FILL_A                                                                                                            \
    "global_store_dwordx4 %7, v[4:7], off offset:32\n" GAP "v_bfe_u32 v4, v5, 16, 1\n v_mov_b32 v7, 0\n v_mov_b32 v6, 0\n v_mov_b32 v5, 0\n" \
    FILL_B                                                                                                            \
    "global_store_dwordx4 %7, v[8:11], off offset:48\n" GAP "v_pk_mul_f32 v[10:11], v[8:9], v[8:9]\n v_mov_b32 v8, 0\n"                \
    "s_waitcnt vmcnt(0)\n"

// 8-byte stores (hipcc assumes no hazard up to 64 bits of data): four global_store_dwordx2 per thread, data overwritten
// by a packed-math write, a v_mov of the upper dword, a VOP3 write of the lower dword, a packed write again
#define G2BODY(GAP)                                                                                                    \
    FILL_A FILL_B                                                                                                     \
    "global_store_dwordx2 %7, v[4:5], off\n" GAP "v_pk_mul_f32 v[4:5], v[4:5], v[4:5]\n"                              \
    "global_store_dwordx2 %7, v[6:7], off offset:8\n" GAP "v_mov_b32 v7, 0\n v_mov_b32 v6, 0\n"                       \
    "global_store_dwordx2 %7, v[8:9], off offset:16\n" GAP "v_bfe_u32 v8, v9, 16, 1\n v_mov_b32 v9, 0\n"              \
    "global_store_dwordx2 %7, v[10:11], off offset:24\n" GAP "v_pk_mul_f32 v[10:11], v[8:9], v[8:9]\n"                \
    "s_waitcnt vmcnt(0)\n"

template <int KIND, int GAPS>
__global__ void probe(float* out, unsigned nbytes) {
    const unsigned long long p = (unsigned long long)out;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffff);
    r[2] = __builtin_amdgcn_readfirstlane(nbytes);
    r[3] = 0x00020000;
    const unsigned voff = (blockIdx.x * blockDim.x + threadIdx.x) * 64u;   // 4 x 16 B per thread
    float* gp = out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 16;
    const float val = 1.0f + threadIdx.x;
    const unsigned s0 = __builtin_amdgcn_readfirstlane(nbytes & 0u), s1 = s0 + 16, s2 = s0 + 32, s3 = s0 + 48;
#define OPERANDS ::"v"(voff), "v"(val), "s"(r), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "v"(gp) : "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "memory"
#define SGPR_SO(G) asm volatile(BODY("%3 offen", "%4 offen", "%5 offen", "%6 offen", G) OPERANDS)
#define IMM_SO(G) asm volatile(BODY("0 offen", "0 offen offset:16", "0 offen offset:32", "0 offen offset:48", G) OPERANDS)
#define GLOBAL(G) asm volatile(GBODY(G) OPERANDS)
    if (KIND == 0 && GAPS == 0) SGPR_SO("");
    if (KIND == 0 && GAPS == 1) SGPR_SO("s_nop 0\n");
    if (KIND == 0 && GAPS == 2) SGPR_SO("s_nop 1\n");
    if (KIND == 1 && GAPS == 0) IMM_SO("");
    if (KIND == 1 && GAPS == 1) IMM_SO("s_nop 0\n");
    if (KIND == 1 && GAPS == 2) IMM_SO("s_nop 1\n");
    if (KIND == 2 && GAPS == 0) GLOBAL("");
    if (KIND == 2 && GAPS == 1) GLOBAL("s_nop 0\n");
    if (KIND == 2 && GAPS == 2) GLOBAL("s_nop 1\n");
    if (KIND == 3 && GAPS == 0) asm volatile(G2BODY("") OPERANDS);
    if (KIND == 3 && GAPS == 1) asm volatile(G2BODY("s_nop 0\n") OPERANDS);
}

template <int KIND, int GAPS>
bool run(float* d, std::vector<float>& h, int blocks, int threads) {
    const size_t n = h.size();
    static const char* kinds[4] = {"buffer_store_dwordx4, SGPR soffset     ", "buffer_store_dwordx4, immediate soffset", "global_store_dwordx4                   ",
                                   "global_store_dwordx2 (8 B per store)   "};
    static const char* writers16[4] = {"v_pk_mul d2:3", "v_mov d3", "v_bfe(VOP3) d0", "v_pk_mul d2:3"};
    static const char* writers8[4] = {"v_pk_mul d0:1", "v_mov d1", "v_bfe(VOP3) d0", "v_pk_mul d0:1"};
    const char** writers = KIND == 3 ? writers8 : writers16;
    long bad[4][4] = {};
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipMemset(d, 0xff, n * 4);
        probe<KIND, GAPS><<<blocks, threads>>>(d, (unsigned)(n * 4));
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return false; }
        (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        for (size_t t = 0; t < (size_t)blocks * threads; ++t) {
            const float want = 1.0f + (t % threads);
            for (int s = 0; s < 4; ++s)
                for (int e = 0; e < (KIND == 3 ? 2 : 4); ++e) bad[s][e] += h[t * 16 + s * (KIND == 3 ? 2 : 4) + e] != want;
        }
    }
    long total = 0;
    printf("%s  %d wait state(s) between the store and the write of its data registers\n", kinds[KIND], GAPS);
    for (int s = 0; s < 4; ++s) {
        printf("    store %d (behind it: %-14s)  wrong dwords [d0 d1 d2 d3] = [%ld %ld %ld %ld] of %zu each\n", s, writers[s], bad[s][0], bad[s][1],
               bad[s][2], bad[s][3], (size_t)blocks * threads * 5);
        total += bad[s][0] + bad[s][1] + bad[s][2] + bad[s][3];
    }
    return total == 0;
}

int main() {
    const int blocks = 256 * 32, threads = 256;
    const size_t n = (size_t)blocks * threads * 16;
    float* d = nullptr;
    if (hipMalloc(&d, n * 4) != hipSuccess) return 2;
    std::vector<float> h(n);
    run<0, 0>(d, h, blocks, threads);
    run<0, 1>(d, h, blocks, threads);
    bool ok = run<0, 2>(d, h, blocks, threads);
    run<1, 0>(d, h, blocks, threads);
    run<1, 1>(d, h, blocks, threads);
    ok &= run<1, 2>(d, h, blocks, threads);
    run<2, 0>(d, h, blocks, threads);
    run<2, 1>(d, h, blocks, threads);
    ok &= run<2, 2>(d, h, blocks, threads);
    const bool ok8 = run<3, 0>(d, h, blocks, threads);
    run<3, 1>(d, h, blocks, threads);
    printf("8-byte stores are %s without a wait state\n", ok8 ? "safe" : "NOT safe");
    printf("two wait states are %s for every form\n", ok ? "enough" : "NOT enough");
    return ok ? 0 : 1;
}
