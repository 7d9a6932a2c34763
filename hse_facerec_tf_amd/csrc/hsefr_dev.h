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
/* Diagnostic builds only (-DHSEFR_PWS_STAMPS): per-wave phase cycle sums of the last split-f16 GEMM launch;
 * HSEFR_ERR_UNSUPPORTED in the shipped library. */
int hsefr_debug_read_stamps(void* host_out, size_t bytes);
int hsefr_debug_clock_probe(unsigned long long* d_out, int blocks, int iters, hsefr_stream_t stream);

#ifdef __cplusplus
}
#endif
