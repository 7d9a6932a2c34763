/* Development-build additions to the libhsefr ABI (compiled with -DHSEFR_DEV only; tools/kbench.py and the tuning
 * scripts use them).  The product library neither declares nor exports any of this: there every knob is a constant. */
#pragma once
#include "../../include/hsefr.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Tuning/debug knobs, process-wide, never needed for correct results.
 * "pw_tile": -1 = choose per layer (default), 0 = 128x128, 1 = 128x64, 2 = 64x64 GEMM tile.
 * "pw_dma":  1 = GEMM tiles staged by LDS-DMA (global_load_lds, default), 0 = through registers.
 * "pw_ablate": timing-only ablations of the GEMM (results are WRONG): bit 0 = no global loads after the first
 *            K-tile, bit 1 = no epilogue stores.  0 = the real kernel (default).
 * "dw_th":   0 = choose per layer (default), >0 = output rows per depthwise strip.
 * "dw_variant": cache policy of the depthwise kernel: bit 0 = nontemporal loads, bit 1 = nontemporal stores.
 * "copy_variant": shape of the hsefr_debug_copy calibration kernel (unroll / nontemporal / grid bits).
 * "c3_impl": 0 = auto (default), 1 = VALU first-conv kernel, 2 = im2col fp32-MFMA first-conv kernel. */
int hsefr_debug_set(const char* key, int value);
/* Calibration: plain float4 device-to-device copy kernel (the practical HBM ceiling on this GPU). */
int hsefr_debug_copy(const void* d_src, void* d_dst, size_t bytes, hsefr_stream_t stream);
/* Calibration: dense fp32-MFMA loop on `blocks` workgroups; d_out[3*b] = shader-clock ticks, d_out[3*b+1] = 100 MHz
 * ticks of workgroup b (clock under fp32-matrix load = ratio * 100 MHz; 4*iters MFMAs of 4096 FLOP per wave). */
/* Diagnostic builds only (-DHSEFR_*_STAMPS): per-wave phase cycle sums of the last launch of the kernel named by `kernel`
 * (round 6: an id, where the buffer's byte count used to pick the kernel); HSEFR_ERR_UNSUPPORTED without the matching define. */
typedef enum hsefr_stamp_kernel {
    HSEFR_STAMPS_PWS = 0,  /* pwconv_f16s.hip: the split-f16 GEMM            [1024][8][8] words */
    HSEFR_STAMPS_STEM = 1, /* stem4_fused.hip / stem5_stream.hip            [512 * 4][10]     */
    HSEFR_STAMPS_PS = 2,   /* pwconv_ps.hip: the pre-split GEMM              [256][12][8]      */
    HSEFR_STAMPS_CD = 3,   /* conv_dma_bf16.hip                              [256][12][8]      */
    HSEFR_STAMPS_C11 = 4,  /* conv1x1_bf16.hip: the register-staged 1x1 GEMM [512 * 4][8]      */
    HSEFR_STAMPS_W4 = 5,   /* conv1x1_w4_bf16.hip: the four-wave 1x1 GEMM    [256][8][8]       */
    HSEFR_STAMPS_W2 = 6,   /* conv3x3_w2_bf16.hip: the four-wave window 3x3  [256][8][8]       */
    HSEFR_STAMPS_W3 = 7,   /* conv3x3_win_bf16.hip: the window 3x3           [256][12][8]      */
    HSEFR_STAMPS_S7 = 8    /* stem7s_stream.hip: ResNet's streaming stem     [512][4][10] (-DHSEFR_S7_STAMPS) */
} hsefr_stamp_kernel;
int hsefr_debug_read_stamps(int kernel, void* host_out, size_t bytes);
int hsefr_debug_clock_probe(unsigned long long* d_out, int blocks, int iters, hsefr_stream_t stream);


/* Round 1's fused stem (conv1 + the whole first block), reachable only through lower_graph(stem_fusion="stem"): kept for A/B timing
 * against the stems the product runs (stem2 / stem3 / stem4 / stem5) -- HSEFR_OP_STEM_F16S plans run on development builds only. */
/* The MobileNet stem in one kernel (graph nodes #30-#49): conv 3x3 stride 2 SAME (3 -> 32) + shift + ReLU6 -> depthwise
 * 3x3 stride 1 SAME + scale + shift + ReLU6 -> pointwise 1x1 (32 -> 64) + shift + act (split-f16 products).
 * x [n,h,w,3]; conv_w TF HWIO [3,3,3,32]; wd [3,3,32]; w_split/descale as for hsefr_pwconv1x1_f16split;
 * y [n,oh,ow,64] with oh = ceil(h/2), ow = ceil(w/2); cpad_t/cpad_l = the conv's top/left padding. */
int hsefr_stem_fused(const float* x, const float* conv_w, const float* conv_shift, const float* wd, const float* dscale,
                     const float* dshift, const void* w_split, const float* descale, const float* pshift, float* y, int n,
                     int h, int w, int cpad_t, int cpad_l, int oh, int ow, int a_log2, int act, hsefr_stream_t stream);

#ifdef __cplusplus
}
#endif
