// Shared helpers for libhsefr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/hsefr.h"
#ifdef HSEFR_DEV
#include "hsefr_dev.h"
#endif

namespace hsefr {

void set_error(const char* fmt, ...);

#define HSEFR_HIP_CHECK(expr)                                                          \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            ::hsefr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                               __FILE__, __LINE__);                                    \
            return HSEFR_ERR_HIP;                                                      \
        }                                                                              \
    } while (0)

#define HSEFR_REQUIRE(cond, code, ...)     \
    do {                                   \
        if (!(cond)) {                     \
            ::hsefr::set_error(__VA_ARGS__); \
            return (code);                 \
        }                                  \
    } while (0)

// Route probe (hsefr_plan_describe): with the probe active on this host thread every launcher runs as usual up to the launch itself --
// its shape checks, its choice of kernel family and template -- but HSEFR_LAUNCH records the kernel's name instead of launching it and no
// HIP call is made: the plan's op -> kernel table comes from the SAME code that launches, on a machine with or without a GPU.
bool route_probe();
void route_record(const void* host_stub, const char* expr);      // the kernel's host-side stub (its symbol carries the template arguments)
#define HSEFR_LAUNCH(kernel, grid, block, lds, stream, ...)                                                     \
    do {                                                                                                        \
        if (::hsefr::route_probe()) ::hsefr::route_record(reinterpret_cast<const void*>(&kernel), #kernel);     \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                 \
    } while (0)

inline int launch_status(const char* what) {
    if (route_probe()) return HSEFR_OK;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return HSEFR_ERR_HIP;
    }
    return HSEFR_OK;
}

// Epilogue activation.  RELU6 restates the graph's Relu -> Minimum(.,6) -> Maximum(.,0)
// chain (nodes #32-34): max(min(max(x,0),6),0) == min(max(x,0),6).
template <int ACT>
__device__ __forceinline__ float apply_act(float v) {
    if (ACT == HSEFR_ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == HSEFR_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    if (ACT == HSEFR_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

__device__ __forceinline__ float apply_act_rt(float v, int act) {
    switch (act) {
        case HSEFR_ACT_RELU: return fmaxf(v, 0.f);
        case HSEFR_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
        case HSEFR_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// XCD-aware, bijective remap of a linear workgroup id: the hardware deals consecutive
// workgroup ids round-robin over the 8 XCDs (each with a private 4 MiB L2), so workgroups
// that share operand rows (GEMM N-tiles of one M-tile, vertically adjacent depthwise
// row-tiles) are given ids that land on ONE XCD.  Speed only -- never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned NX = 8;
    const unsigned q = nwg / NX, r = nwg % NX;
    const unsigned xcd = bid % NX, idx = bid / NX;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

typedef float hsefr_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hsefr_u32x4 __attribute__((ext_vector_type(4)));

// Raw buffer resource over [p, p + bytes): loads beyond it return 0, stores beyond it are dropped -- that is how tail
// tiles (rows >= M) are handled, with no clamping and no branches.  One VGPR byte offset per thread is constant for the
// whole kernel; everything that changes (tile, staging pass, K-tile) is uniform and travels in the SGPR offset.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long long bytes) {
    const unsigned n = bytes <= 0 ? 0u : (bytes > 0xffffffffll ? 0xffffffffu : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);
}
// gfx950 store-data hazard (DESIGN.md lesson 14, tools/store_hazard_probe.hip): a vector write of a 16-byte store's
// data VGPRs one instruction behind it (two for global / immediate-offset stores) reaches memory instead of the stored
// value, and hipcc guarantees one wait state fewer than the hardware needs.  An instruction behind every buffer store
// closes the SGPR-offset form; tools/isa_lint.py checks the linked library for every form.
__device__ __forceinline__ void hsefr_store_guard() { asm volatile("s_nop 0"); }
__device__ __forceinline__ hsefr_f32x4 bload16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(hsefr_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// The same store with its wait state welded on (one asm statement: nothing can be scheduled in between).  hipcc does not
// see the store, so its own vmcnt arithmetic is off by it -- only for kernels whose vector-memory waits are all explicit.
__device__ __forceinline__ void bstore16_welded(hsefr_f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(r), "s"(soff) : "memory");
}
// The same with the non-temporal hint: for a layer's output that nothing re-reads before the next launch streams it in.  Measured per
// kernel in the network (round 5): the epilogue GEMMs' split rows -2.5 us per layer; the fused blocks' and the pointwise GEMMs' outputs
// must NOT have it (the depthwise layers behind them read more slowly: +7 / +3 us), nor the bf16 ResNet kernels' (-5 % on the forward).
__device__ __forceinline__ void bstore16_welded_nt(hsefr_f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(r), "s"(soff) : "memory");
}
__device__ __forceinline__ void bstore16(hsefr_f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(hsefr_u32x4, v), r, voff, soff, 0);
    hsefr_store_guard();
}

// fp32 -> bf16, round to nearest even, on gfx950's own instruction (v_cvt_pk_bf16_f32: two values per instruction).  Until round 5
// every bf16 epilogue rounded with integer arithmetic on the bits -- (u + 0x7FFF + ((u >> 16) & 1)) >> 16: four vector
// instructions per value, and the bf16 convolutions turned out to be EPILOGUE-bound where their K loop is short (the 1x1
// "increase" layers: ~600 vector instructions per thread and 128 x 128 tile).  Same result for every finite input (NaNs stay NaNs
// here; the integer form could turn one into an infinity: MI355X guide, correctness boundaries).
typedef __bf16 hsefr_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hsefr_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned hsefr_bf16_bits(float f) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned hsefr_pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(hsefr_f32x2{lo, hi}, hsefr_bf16x2));
}

// Exact unsigned division by a launch-invariant divisor d >= 2, for EVERY 32-bit numerator (the "add" form of division by multiplication:
// q0 = mulhi(x, m), q = (((x - q0) >> 1) + q0) >> sh with m = floor(2^32 (2^s - d) / d) + 1, s = ceil(log2 d), sh = s - 1).  Four vector
// instructions; the plain mulhi-by-ceil(2^32 / d) shortcut is exact only while x d < 2^32, which a large batch of large maps exceeds.
struct hsefr_udiv { unsigned m, sh; };
inline hsefr_udiv hsefr_udiv_make(unsigned d) {       // d >= 2
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    hsefr_udiv r;
    r.m = (unsigned)((((1ull << s) - d) << 32) / d + 1);
    r.sh = s - 1;
    return r;
}
__device__ __forceinline__ unsigned hsefr_udiv_do(unsigned x, hsefr_udiv k) {
    const unsigned q0 = __umulhi(x, k.m);
    return (((x - q0) >> 1) + q0) >> k.sh;
}

// Sweep direction.  Every kernel walks its output (and so its input) in one address order; consecutive layers sweep in
// OPPOSITE orders, so that a layer starts on the bytes its producer wrote last -- the ones still in the 256 MiB
// Infinity Cache -- instead of chasing an LRU that evicts every line just before it is needed (a 302 MB activation
// swept twice in the same order never hits).  Set by the engine per op (op index parity); 0 for direct ABI calls.
__device__ __forceinline__ unsigned xcd_remap_dir(unsigned bid, unsigned nwg, int reverse) {
    const unsigned lt = xcd_remap(bid, nwg);
    return reverse ? nwg - 1u - lt : lt;
}
int sweep_reverse();          // per host thread (thread_local in engine.hip): engines on different threads do not interfere
void set_sweep_reverse(int v);

// Tuning knobs exist in DEVELOPMENT builds only (build.sh with HSEFR_DEV=1 -> -DHSEFR_DEV: hsefr_debug_set, devtools.hip,
// tools/kbench.py).  In the product library every knob is a compile-time constant: the measured-best value, no
// process-global mutable state, and the variants behind the other values are not even compiled.
#ifdef HSEFR_DEV
#define HSEFR_KNOB(name, value) int name = value
#else
#define HSEFR_KNOB(name, value) constexpr int name = value
#endif

// --- launchers implemented in the .hip files (same ones the engine calls) -------------
int launch_conv_c3(const float* x, const float* wgt, const float* shift, float* y, int n, int h, int w,
                   int kh, int kw, int stride, int pad_t, int pad_l, int oh, int ow, int cout, int act,
                   hipStream_t s);
int launch_dwconv3x3(const float* x, const float* wgt, const float* scale, const float* shift, float* y,
                     int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh, int ow, int act,
                     hipStream_t s);
int launch_dwconv3x3_split(const float* x, const float* wgt, const float* scale, const float* shift, void* y_split,
                           int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh, int ow, int act,
                           int a_log2, hipStream_t s);
int launch_pwconv_f32(const float* x, const float* wgt_t, const float* shift, float* y, long long m, int k,
                      int cout, int act, hipStream_t s);
int launch_pwconv_f16s(const float* x, const void* wsplit, const float* descale, const float* shift, float* y,
                       long long m, int k, int cout, int a_log2, int act, hipStream_t s);
// pre-split activations (csrc/pwconv_ps.hip): xs = split rows [m][k/32][hi 32 x f16 | lo 32 x f16], scaled by 2^a_log2 (folded in descale)
bool pwconv_ps_supported(long long m, int k, int cout);
bool pwconv_ps_gap_supported(long long m, int k, int cout, int map_hw);
int launch_pwconv_ps_gap(const void* xs, const void* wsplit, const float* descale, const float* shift, float* y, long long m, int k, int cout,
                         int act, int map_hw, hipStream_t s);
bool pwconv_ps_dw_supported(long long m, int k, int cout, int map_w, int map_hw, int dw_stride);
int launch_pwconv_ps_dw(const void* xs, const void* wsplit, const float* descale, const float* shift, const float* dwc, void* ys, long long m,
                        int k, int cout, int act, int map_w, int map_hw, int dw_stride, int out_log2, hipStream_t s);
int launch_pwconv_ps(const void* xs, const void* wsplit, const float* descale, const float* shift, float* y, long long m, int k,
                     int cout, int act, hipStream_t s);
int launch_gap(const float* x, float* y, int n, int hw, int c, hipStream_t s);
int launch_dense(const float* x, const float* wgt, const float* bias, float* y, int n, int k, int cout,
                 int act, hipStream_t s);
int launch_softmax(const float* x, float* y, int n, int c, hipStream_t s);
int launch_heads_fused(const float* x, const float* w1, const float* b1, const float* wa, const float* ba, const float* wg, const float* bg,
                       float* hidden, float* logits, float* age_probs, float* gender, int n, int k, int a, hipStream_t s);
int launch_l2_normalize(const float* x, float* y, int n, int d, hipStream_t s);
int launch_nn1(const float* q, const float* g, int nq, int ng, int d, int* nn_index, float* nn_dist2,
               hipStream_t s);
long long nn1_fallbacks();

int launch_conv_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y,
                     int n, int h, int w, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t,
                     int pad_l, int act, hipStream_t s);
int launch_conv1x1_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y,
                        long long P, int K, int cout, int act, hipStream_t s);
int launch_conv1x1_sres_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y, int n, int oh,
                             int ow, int K, int cout, int stride, int h2, int w2, int act, hipStream_t s);
bool conv1x1_bf16_enabled(bool has_res, int k, int cout);
int launch_conv1x1_proj_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* x2, const void* wt2,
                             const float* scale2, const float* shift2, void* y, int n, int oh, int ow, int K, int cout, int k2, int stride,
                             int h2, int w2, int act, hipStream_t s);
int launch_stem7x7_bf16(const float* x, const void* wt, const float* scale, const float* shift, void* y, int n, int h,
                        int w, int oh, int ow, int act, hipStream_t s);
int mtcnn_post_capacity();
int launch_mtcnn_stage1_level(const float* prob, const float* reg, int w, int h, double scale, float thr, double* found, int* counters,
                              hipStream_t s);
int launch_mtcnn_stage1_finish(const double* found, int* counters, double* boxes, int* tab, int img_w, int img_h, hipStream_t s);
int launch_mtcnn_stage23_finish(int stage, const double* boxes_in, int n, const float* prob, const float* reg, const float* pts, float thr,
                                double* boxes_out, int* tab_out, float* points_out, int* counters, int img_w, int img_h, hipStream_t s);
int launch_mtcnn_nms(const double* boxes, int n, double thr, int use_min, int* keep, int* n_keep, hipStream_t s);
bool conv3x3_win_forced();
bool conv3x3_win_bf16_supported(long long n, int h, int w, int c, int cout);
int launch_conv3x3_win_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y, int n, int h,
                            int w, int c, int cout, int act, hipStream_t s);
bool conv3x3_w2_forced();
bool conv3x3_w2_bf16_supported(long long n, int h, int w, int c, int cout);
bool conv3x3_w2_bf16_preferred(int h, int w, int cout);
int launch_conv3x3_w2_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y, int n, int h,
                           int w, int c, int cout, int act, hipStream_t s);
bool conv1x1_w4_forced();
bool conv1x1_w4_bf16_supported(long long n, int h, int w, int c, int oh, int ow, int cout, int stride);
bool conv1x1_w4_bf16_preferred(long long pixels, int c, int cout, bool has_res);
bool conv1x1_w4_proj_preferred(long long pixels, int c, int c2, int cout);
int launch_conv1x1_w4_proj_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* x2, const void* wt2,
                                const float* scale2, const float* shift2, void* y, int n, int oh, int ow, int c, int cout, int c2, int stride2,
                                int h2, int w2, int act, hipStream_t s);
int launch_conv1x1_w4_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y, int n, int h,
                           int w, int c, int oh, int ow, int cout, int stride, int act, hipStream_t s);
bool conv1x1_pair_bf16_shape_supported(int c, int cout1, int cout2, int c2);
bool conv1x1_pair_bf16_supported(long long pixels, int c, int cout1, int cout2, int c2);
int launch_conv1x1_pair_bf16(const void* x, const void* w1, const float* scale1, const float* shift1, const void* res, const void* x2,
                             const void* wp, const float* scale_p, const float* shift_p, void* y1, const void* w2, const float* scale2,
                             const float* shift2, void* y2, long long pixels, int c, int cout1, int cout2, int c2, int act1, int act2,
                             hipStream_t s, int y1_stride = 1, int h = 0, int w = 0);
bool conv_dma_forced();
bool conv_dma_bf16_supported(long long n, int h, int w, int c, int oh, int ow, int cout, int kh, int kw);
int launch_conv_dma_bf16(const void* x, const void* wt, const float* scale, const float* shift, const void* res, void* y, int n, int h,
                         int w, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, int act, hipStream_t s);
// wfrag: the weight image in the streaming kernel's fragment order (stem7s_reorder_weights; STEM7S_WFRAG_BYTES), or null -- then the kernel
// re-orders wt itself, in every workgroup's prologue.  The engine makes the image once per plan.
int launch_stem7x7_pool_bf16(const float* x, const void* wt, const float* scale, const float* shift, void* y, int n, int h, int w,
                             int ph, int pw, int pool_pad_t, int pool_pad_l, hipStream_t s, const void* wfrag = nullptr);
bool stem7s_stream_supported(long long n, int h, int w, int ph, int pw);
int launch_stem7s_stream(const float* x, const void* wt, const float* scale, const float* shift, void* y, int n, int h, int w, int ph, int pw,
                         int pool_pad_t, int pool_pad_l, hipStream_t s, const void* wfrag = nullptr);
constexpr size_t STEM7S_WFRAG_BYTES = 4 * 7 * 64 * 16;
int stem7s_reorder_weights(const void* wt, void* wfrag, hipStream_t s);
int launch_maxpool3x3s2_bf16(const void* x, void* y, int n, int h, int w, int c, int oh, int ow, int pad_t, int pad_l,
                             hipStream_t s);
int launch_gap_bf16(const void* x, float* y, int n, int hw, int c, hipStream_t s);

int launch_pil_resize(const unsigned char* in, unsigned char* tmp, float* out, int n, int H, int W, int oh, int ow, const int* xmin,
                      const int* xcnt, const int* xcoef, int xk, const int* ymin, const int* ycnt, const int* ycoef, int yk,
                      int mode, const double* mean, hipStream_t s);
int launch_cv_resize(const unsigned char* in, float* out, int n, int H, int W, int oh, int ow, const int* x0, const int* x1,
                     const int* wx1, const int* y0, const int* y1, const int* wy1, int mode, const double* mean, hipStream_t s);

int launch_area_level(const unsigned char* src, float* dst, int sh, int sw, int dh, int dw, hipStream_t s);
int launch_area_crops(const unsigned char* src, const int* boxes, float* dst, int sh, int sw, int n, int size, hipStream_t s);
int launch_pairwise_dist(const float* x, const float* y, int n, int m, int d, float* out, hipStream_t s);

bool dwpw_fused_supported(int c, int cout, int stride, int act_dw, int act_pw);
int launch_dwpw_fused(const float* x, const float* wd, const float* dscale, const float* dshift, const float* wp_t,
                      const float* pshift, float* y, int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh,
                      int ow, int cout, int act_dw, int act_pw, hipStream_t s);

int launch_conv2d_direct(const float* x, const float* w, const float* bias, const float* alpha, float* y, int n, int h, int wd, int c,
                         int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, hipStream_t s);
int launch_conv_f32_mfma(const float* x, const float* w, const float* scale, const float* shift, const float* res, float* y, int n, int h,
                         int wd, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, int act, hipStream_t s,
                         int res_stride = 0, int res_h = 0, int res_w = 0);
bool conv_f32_mfma_supported(int c, int cout);
int launch_conv2d_f32(const float* x, const float* w, const float* scale, const float* shift, const float* res, float* y, int n, int h,
                      int wd, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, int act, hipStream_t s);
int launch_maxpool_f32(const float* x, float* y, int n, int h, int w, int c, int oh, int ow, int k, int stride, int pad_t, int pad_l,
                       hipStream_t s);

int launch_dwpw_f16s(const float* x, const float* wd, const float* dscale, const float* dshift, const void* wsplit,
                     const float* descale, const float* pshift, float* y, int n, int h, int w, int c, int stride, int pad_t,
                     int pad_l, int oh, int ow, int cout, int a_log2, int act, hipStream_t s);
bool dwpw_f16s_supported(int c, int cout, int stride);
int launch_stem_fused(const float* x, const float* cw, const float* cshift, const float* wd, const float* dscale,
                      const float* dshift, const void* wsplit, const float* descale, const float* pshift, float* y, int n,
                      int h, int w, int cpad_t, int cpad_l, int oh, int ow, int a_log2, int act, hipStream_t s);
int launch_stem2_fused(const float* x, const float* cw, const float* cshift, const float* wd1, const float* d1scale,
                       const float* d1shift, const void* wsplit, const float* descale, const float* pshift, const float* wd2,
                       const float* d2scale, const float* d2shift, float* y, int n, int h, int w, int cpad_t, int cpad_l,
                       int h1, int w1, int pad_t2, int pad_l2, int oh2, int ow2, int a_log2, int act, hipStream_t s);
int launch_stem3_fused(const float* x, const void* cw_split, const float* cdescale, const float* cshift, const float* wd1,
                       const float* d1scale, const float* d1shift, const void* wsplit, const float* descale, const float* pshift,
                       const float* wd2, const float* d2scale, const float* d2shift, float* y, int* overflow, int n, int h, int w,
                       int cpad_t, int cpad_l, int h1, int w1, int pad_t2, int pad_l2, int oh2, int ow2, int in_log2, int a_log2,
                       int act, hipStream_t s);
bool stem3_fused_supported(int cin, int c1, int c2, int conv_stride, int dw1_stride, int dw2_stride, int kh, int kw);
int launch_stem4_fused(const void* x, int x_is_u8, const void* cw4, const float* cdescale, const float* cshift, const float* wd1,
                       const float* d1scale, const float* d1shift, const void* wsplit, const float* descale, const float* pshift,
                       const float* wd2, const float* d2scale, const float* d2shift, float* y, int* overflow, int n, int h, int w,
                       int in_log2, int a_log2, int act, hipStream_t s);
bool stem4_fused_supported(int cin, int c1, int c2, int conv_stride, int dw1_stride, int dw2_stride, int kh, int kw, int h, int w);
int launch_stem5_stream(const void* x, int x_is_u8, const void* cw4, const float* cdescale, const float* cshift, const float* wd1,
                        const float* d1scale, const float* d1shift, const void* wsplit, const float* descale, const float* pshift,
                        const float* wd2, const float* d2scale, const float* d2shift, float* y, int* overflow, int n, int h, int w,
                        int in_log2, int a_log2, int act, hipStream_t s);
bool stem5_stream_supported(int cin, int c1, int c2, int conv_stride, int dw1_stride, int dw2_stride, int kh, int kw, int h, int w);
bool stem2_fused_supported(int cin, int c1, int c2, int conv_stride, int dw1_stride, int dw2_stride, int kh, int kw);
bool stem_fused_supported(int cin, int cmid, int cout, int conv_stride, int dw_stride, int kh, int kw);
int read_pws_stamps(void* host_out, size_t bytes);
int read_stem_stamps(void* host_out, size_t bytes);
#ifdef HSEFR_STEM_STAMPS
// Diagnostic builds: per-wave s_memtime sums of kernel phases, [512 workgroups][4 waves][8 phases, lifetime, count].
unsigned long long* stamp_buffer(hipStream_t s);
#define STEM_STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); st[i] += _t - tprev; tprev = _t; } while (0)
#define STEM_STAMP_DECL unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tprev = __builtin_amdgcn_s_memtime(); \
    const unsigned long long tstart = tprev; unsigned npatch = 0
#define STEM_STAMP_COUNT ++npatch
#define STEM_STAMP_FLUSH(buf, lane, wave) do { if ((lane) == 0 && (buf) && blockIdx.x < 512) { unsigned long long* o = (buf) + (blockIdx.x * 4 + (wave)) * 10; \
    for (int i_ = 0; i_ < 8; ++i_) o[i_] = st[i_]; o[8] = __builtin_amdgcn_s_memtime() - tstart; o[9] = npatch; } } while (0)
#else
#define STEM_STAMP(i) do { } while (0)
#define STEM_STAMP_DECL do { } while (0)
#define STEM_STAMP_COUNT do { } while (0)
#define STEM_STAMP_FLUSH(buf, lane, wave) do { } while (0)
#endif
#ifdef HSEFR_DEV
void set_stem4_grid(int v);
void set_stem5_grid(int v);
void set_stem5_segs(int v);
void set_c11(int v);
void set_c11_tile(int v);
void set_c11_bres(int v);
void set_c11_adv(int v);
int read_c11_stamps(void* host_out, size_t bytes);
void set_dwpws_tw(int v);
void set_dwpws_bn(int v);
void set_pw_tile(int v);
void set_pws_tile(int v);
void set_ps_mb(int v);
void set_ps_grid(int v);
void set_psdw_mode(int v);
void set_cd_rb(int v);
void set_w3_off(int v);
void set_w2_off(int v);
void set_w4_off(int v);
void set_w4_bres(int v);
void set_pair_off(int v);
void set_stem7s(int v);
void set_pair_ablate(int v);
void set_pair_nt(int v);
void set_nn1_y_mb(int v);
int read_w4_stamps(void* host_out, size_t bytes);
int read_s7_stamps(void* host_out, size_t bytes);
int read_w2_stamps(void* host_out, size_t bytes);
int read_w3_stamps(void* host_out, size_t bytes);
int read_cd_stamps(void* host_out, size_t bytes);
void set_cd_off(int v);
int read_ps_stamps(void* host_out, size_t bytes);
void set_pw_ablate(int v);
void set_pw_dma(int v);
void set_dw_th(int v);
void set_dw_variant(int v);
void set_dw_look(int v);
void set_dw_look2(int v);
void set_copy_variant(int v);
void set_clock_mode(int v);
void set_c3_impl(int v);
int launch_clock_probe(unsigned long long* out, int blocks, int iters, hipStream_t s);
int launch_copy(const void* src, void* dst, size_t bytes, hipStream_t s);
#endif

}  // namespace hsefr
