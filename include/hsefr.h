/*
 * hsefr.h -- C ABI of libhsefr.so, the MI355X (gfx950) forward-pass engine that replaces
 * the TensorFlow `Session.run` call of av-savchenko/HSE_FaceRec_tf's feature-extract path.
 *
 * What it replaces in the reference (the reference has a Python-object boundary, no FFI;
 * this is the thin layer a binding puts under those objects -- see INTEGRATION.md):
 *   - tf.import_graph_def + tf.Session(graph)   facerec_test.py:41-48,58 ; facial_analysis.py:55-58
 *       -> hsefr_engine_create()  (the caller lowers the frozen GraphDef to a "plan" first)
 *   - tf_sess.run(out, {in: x})                 facerec_test.py:117-120 (features)
 *     sess.run([age, gender, feats], {in: x})   facial_analysis.py:109   (three fetches)
 *       -> hsefr_engine_forward()
 *   - tf_sess.close() / sess.close()            facerec_test.py:124-125 ; facial_analysis.py:73-74
 *       -> hsefr_engine_destroy()
 *   - sklearn normalize + KNeighborsClassifier(1).kneighbors   facerec_test.py:401,200-207,422
 *       -> hsefr_l2_normalize() + hsefr_nn1()
 *   - misc.imresize / cv2.resize + BGR + mean   facerec_test.py:93-106 ; facial_analysis.py:95-107
 *       -> hsefr_preprocess_pil_u8() / hsefr_preprocess_cv_u8()
 *
 * Conventions: extern "C", plain C types; every function returns 0 on success or a negative
 * hsefr_status (never throws); hsefr_last_error_string() describes the last failure on the
 * calling thread.  ALL data pointers are DEVICE pointers owned by the caller (e.g. torch
 * `tensor.data_ptr()`); launches are asynchronous on the `stream` argument (a hipStream_t
 * passed as void*; NULL = the default stream).  An engine owns only its weights and its
 * activation workspace; it is re-entrant per engine (one forward at a time per engine),
 * starts no threads and never synchronises the device.  Activations are NHWC, fp32 unless
 * stated.  Built for gfx950 only.
 */
#ifndef HSEFR_H
#define HSEFR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSEFR_VERSION 141 /* 0.1.4: round-6 ABI (added: hsefr_plan_op.flags with HSEFR_OPF_PAIR_NEXT / HSEFR_OPF_HEADS, hsefr_conv1x1_pair_bf16, hsefr_heads_fused,
                             hsefr_plan_validate, hsefr_plan_describe, hsefr_nn1_fallbacks; 141: HSEFR_OPF_OUT_SUB2, hsefr_conv1x1_sres_bf16, hsefr_conv1x1_pair_sub2_bf16, the
                             strided-residual word of HSEFR_OP_CONV_BF16; removed from the product library: hsefr_stem_fused / HSEFR_OP_STEM_F16S (development builds only);
                             130 = round 5, 120 = round 4, 110 = round 3, 100 = round 1-2) */

typedef enum hsefr_status {
    HSEFR_OK = 0,
    HSEFR_ERR_INVALID = -1,     /* bad argument / malformed plan            */
    HSEFR_ERR_UNSUPPORTED = -2, /* shape the kernels do not cover            */
    HSEFR_ERR_HIP = -3,         /* a HIP runtime call failed                 */
    HSEFR_ERR_NOMEM = -4,       /* device or host allocation failed          */
    HSEFR_ERR_SHAPE = -5        /* n > max_batch, etc. (ValueError in Python) */
} hsefr_status;

/* fused activation applied in a kernel's epilogue (graph pattern Relu -> Minimum(.,6) ->
 * Maximum(.,0), nodes #32-34 of the reference's frozen MobileNet, == RELU6) */
typedef enum hsefr_act {
    HSEFR_ACT_NONE = 0,
    HSEFR_ACT_RELU = 1,
    HSEFR_ACT_RELU6 = 2,
    HSEFR_ACT_SIGMOID = 3
} hsefr_act;

typedef void* hsefr_stream_t; /* hipStream_t */

int hsefr_version(void);
const char* hsefr_last_error_string(void);
/* Tuning knobs and calibration kernels (hsefr_debug_*) are NOT part of this ABI: they exist only in development builds
 * of the library (hse_facerec_tf_amd/csrc/hsefr_dev.h, build.sh with HSEFR_DEV=1). */

/* ------------------------------------------------------------------------------------ */
/* Plan: the lowered frozen graph handed to hsefr_engine_create (host memory).           */
/* Layout: hsefr_plan_header | hsefr_plan_buffer[n_buffers] | hsefr_plan_op[n_ops] | blob */
/* All offsets in ops are BYTE offsets into the blob; HSEFR_NO_OFFSET = absent.          */
/* ------------------------------------------------------------------------------------ */
#define HSEFR_PLAN_MAGIC 0x314c505246455348ull /* "HSEFRPL1" little-endian */
#define HSEFR_NO_OFFSET 0xffffffffffffffffull
#define HSEFR_BUF_INPUT (-1) /* op reads the caller's d_input */
#define HSEFR_BUF_NONE (-2)

typedef enum hsefr_op_kind {
    HSEFR_OP_CONV_C3 = 1,      /* dense KxK conv, Cin==3 (MobileNet conv1 3x3/2, ResNet stem 7x7/2), fp32 in */
    HSEFR_OP_DWCONV3X3 = 2,    /* depthwise 3x3 stride 1|2 + scale + shift + act                          */
    HSEFR_OP_PWCONV_F32 = 3,   /* 1x1 conv as fp32-MFMA GEMM + shift + act                                */
    HSEFR_OP_GAP = 4,          /* mean over H,W                                                           */
    HSEFR_OP_DENSE = 5,        /* x[N,K] . W[K,Cout] + b, act                                             */
    HSEFR_OP_SOFTMAX = 6,      /* row softmax                                                             */
    HSEFR_OP_CONV_BF16 = 7,    /* KxK conv (1x1/3x3, stride 1|2) as bf16-MFMA implicit GEMM, fp32 acc,
                                  + shift (+ residual) + act; bf16 activations.  w2_off set: + projected shortcut
                                  (hsefr_conv1x1_proj_bf16); residual, no w2_off, reserved != 0: the residual is a stride view
                                  of a larger map (hsefr_conv1x1_sres_bf16: reserved = stride << 12 | h2 << 14 | w2 << 23)  */
    HSEFR_OP_MAXPOOL_BF16 = 8, /* 3x3/2 max-pool, bf16                                                    */
    HSEFR_OP_GAP_BF16 = 9,     /* mean over H,W of bf16 activations -> fp32                               */
    HSEFR_OP_STEM7X7_BF16 = 10,/* 7x7/2 pad-3 conv over the fp32 3-channel image -> 64 ch bf16 (+scale+shift+ReLU) */
    HSEFR_OP_DWPW_F32 = 11,    /* fused depthwise 3x3 (+scale+shift+ReLU6) -> pointwise 1x1 (+shift+ReLU6), cin 32|64,
                                  cout 64|128: the depthwise result never leaves the CU                            */
    HSEFR_OP_PWCONV_F16S = 12, /* 1x1 conv + shift + act, fp32 in/out, products on the f16 MFMA from a two-term split of
                                  both operands (fp32-grade, csrc/pwconv_f16s.hip); input bounded: |x| * 2^a_log2 < 32768.
                                  w_off = split rows, scale_off = descale, shift_off = shift, `reserved` = a_log2   */
    HSEFR_OP_DWPW_F16S = 13,   /* fused depthwise 3x3 (+scale+shift+ReLU6) -> pointwise 1x1 (+shift+act) for any cin % 32,
                                  cout % 64, pointwise products as in PWCONV_F16S (csrc/dwpw_f16s.hip).  w_off/scale_off/
                                  shift_off = depthwise; w2_off = split rows; shift2_off = [2][cout]: descale, then shift;
                                  `reserved` = a_log2 (the depthwise result is in [0,6]: 12)                         */
    HSEFR_OP_STEM_F16S = 14,   /* DEVELOPMENT BUILDS ONLY since round 6 (the product library answers HSEFR_ERR_UNSUPPORTED; no default lowering emits it).
                                  The whole MobileNet stem (csrc/stem_fused.hip): conv 3x3/2 3->32 + shift + ReLU6 ->
                                  depthwise 3x3/1 + scale + shift + ReLU6 -> pointwise 32->64 + shift + act.  h,w,cin = the
                                  image, oh,ow,cout = the block output, pad_t/pad_l = the conv's; w_off = fp32 pack
                                  [conv HWIO 864 | conv shift 32 | dw 3x3x32 288 | dw scale 32 | dw shift 32];
                                  w2_off = split rows [64][64 f16]; shift2_off = [2][64] descale, shift; reserved = a_log2 */
    HSEFR_OP_CONV_F32 = 18,    /* general KxK conv, stride, zero padding, fp32 NHWC in/out, exact fp32 FMA; epilogue as CONV_BF16
                                  (per-channel scale + shift, optional residual buffer, act): the fp32-grade mode of
                                  ResNet-style graphs (csrc/smallnet.hip).  w_off = TF HWIO kernel fp32, cout % 4 == 0  */
    HSEFR_OP_MAXPOOL_F32 = 19, /* k x k / stride max-pool with windows clipped to the image, fp32 (kh = kw = k)          */
    HSEFR_OP_STEM7X7_POOL_BF16 = 20, /* STEM7X7_BF16 (act must be ReLU) + the 3x3 / 2 max-pool behind it in one kernel
                                  (csrc/stem7s_stream.hip; csrc/stem7x7_pool.hip beyond 32-bit offsets): fp32 image in, POOLED bf16 map out (oh, ow = pooled size);
                                  reserved = pool_pad_t | pool_pad_l << 4, each 0 or 1; blob operands as STEM7X7_BF16    */
    HSEFR_OP_PWCONV_PS_DW = 21, /* PWCONV_PS + the NEXT block's depthwise 3x3 (stride 1, SAME) + scale + shift + ReLU6 in the GEMM's epilogue,
                                  output = that depthwise layer's split rows (csrc/pwconv_ps.hip): h * w <= 288; stride = the depthwise's
                                  (1, or 2 on 12 x 12 / 14 x 14 maps with pad 0), oh / ow = its output size.
                                  w2_off = [11][cout] fp32: taps 0..8, scale * 2^out_log2, shift * 2^out_log2;
                                  reserved = a_log2 | out_log2 << 8                                                         */
    HSEFR_OP_PWCONV_PS_GAP = 22, /* PWCONV_PS + the global average pool behind it in the GEMM's epilogue: output [1,1,cout] fp32 per image
                                  (33 <= h * w <= 288); blob operands as PWCONV_PS, reserved = a_log2            */
    HSEFR_OP_STEM3_F16S = 17,  /* STEM2_F16S for an input with a DECLARED BOUND |x| < 2^(15 - in_log2) (csrc/stem3_fused.hip): conv1's
                                  products are formed on the f16 MFMA from two-term splits like the pointwise layers'.
                                  w_off = the STEM2 fp32 pack (1952 floats) | conv1 split rows [32][64 f16] (1024 floats) |
                                  conv1 descale [32] | the conv kernel in the two-step K layout of csrc/stem4_fused.hip (2048 floats)
                                  | the same channel-reversed for uint8 RGB input (2048) | its four mean-folded shift vectors
                                  [4][32] | its descale [32] = 7264 floats; `reserved` = a_log2 | (in_log2 + 64) << 8 |
                                  (uint8 constants valid ? 1 << 16 : 0); the rest as STEM2_F16S.  Inputs whose edges are
                                  multiples of 4 run stem4_fused.hip, the others stem3_fused.hip.
                                  An input value outside the bound raises the engine's overflow flag (see
                                  hsefr_engine_input_overflow) -- the results of that forward are then meaningless.   */
    HSEFR_OP_PWCONV_PS = 16,   /* PWCONV_F16S whose INPUT buffer holds pre-split activations ("split rows", written by a
                                  DWCONV3X3 op with `reserved` = a_log2 > 0): both GEMM operands go to LDS by DMA
                                  (csrc/pwconv_ps.hip); operands as PWCONV_F16S; k % 32 == 0, cout % 128 == 0            */
    HSEFR_OP_STEM2_F16S = 15   /* the stem plus the depthwise of block 2 (csrc/stem2_fused.hip): ... -> pointwise 32->64 +
                                  shift + ReLU6 -> depthwise 3x3/2 + scale + shift + act.  h,w,cin = the image, oh,ow,cout =
                                  the stride-2 depthwise output (64 ch); pad_t/pad_l = conv1's; kh (sic) low byte = 3,
                                  w_off = fp32 pack [conv HWIO 864 | conv shift 32 | dw1 288 | dw1 scale 32 | dw1 shift 32 |
                                  dw2 3x3x64 576 | dw2 scale 64 | dw2 shift 64]; w2_off / shift2_off / reserved as STEM_F16S;
                                  res_buf unused; stride = 2; `kw` = 3 + 16*pad_t2 + 32*pad_l2 (depthwise-2 padding)   */
} hsefr_op_kind;

typedef enum hsefr_output_slot {
    HSEFR_OUT_FEATURES = 0, /* facerec_test.py:120 output_tensor / 'global_pooling/Mean:0' */
    HSEFR_OUT_AGE = 1,      /* 'age_pred/Softmax:0'                                          */
    HSEFR_OUT_GENDER = 2,   /* 'gender_pred/Sigmoid:0'                                       */
    HSEFR_N_OUTPUT_SLOTS = 3
} hsefr_output_slot;

typedef struct hsefr_plan_header {
    uint64_t magic;
    uint32_t version; /* 2 */
    uint32_t n_buffers;
    uint32_t n_ops;
    uint32_t in_h, in_w, in_c;
    int32_t out_buffer[HSEFR_N_OUTPUT_SLOTS]; /* buffer id per output slot, HSEFR_BUF_NONE if absent */
    uint32_t out_elems[HSEFR_N_OUTPUT_SLOTS]; /* elements per image of that output                   */
    uint64_t blob_bytes;
} hsefr_plan_header;

typedef struct hsefr_plan_buffer {
    uint64_t elems_per_image;
    uint32_t elem_bytes; /* 4 = fp32, 2 = bf16 */
    uint32_t reserved;
} hsefr_plan_buffer;

/* Launch-level fusions the engine applies between CONSECUTIVE ops (round 6).  The ops stay in the plan as they are -- buffers, liveness,
 * per-layer tensors and the CPU plan checker do not change -- and the flagged op's launch also computes the ops behind it, which are then
 * skipped.  hsefr_engine_create validates the pattern; a forward that does not need the covered ops (or an all-layers forward asked to
 * keep their tensors: every tensor is still written) runs the flagged op alone. */
typedef enum hsefr_op_flags {
    HSEFR_OPF_PAIR_NEXT = 1, /* CONV_BF16 1x1 (+ residual | + projected shortcut) whose output the NEXT op, a CONV_BF16 1x1 at the same
                                pixels, reads: both in one launch, the first output stored and chained through registers into the second
                                product (csrc/conv1x1_pair_bf16.hip): ResNet-50's increase -> next reduce in the 56-pixel stage          */
    HSEFR_OPF_HEADS = 2,     /* DENSE k -> 256 + ReLU followed by DENSE 256 -> A (<= 128) + bias, SOFTMAX over it, and DENSE 256 -> 1 +
                                sigmoid: the age / gender heads of facial_analysis.py:109 in one launch (csrc/pool_dense.hip)         */
    HSEFR_OPF_OUT_SUB2 = 4   /* with PAIR_NEXT only: the flagged op's OWN output is stored at the pixels with even row and column only --
                                oh = (h + 1) / 2, ow = (w + 1) / 2 describe the stored map -- while the covered op reads all h x w pixels
                                from registers: the tensor's only other reader takes every second pixel of it (the shortcut of a ResNet
                                stage's last block, lowering.compact_pair_outputs).  Such a pair has no two-launch form: a forward that
                                cannot run it as one launch fails                                                                      */
} hsefr_op_flags;

typedef struct hsefr_plan_op {
    uint32_t kind; /* hsefr_op_kind */
    uint32_t act;  /* hsefr_act     */
    int32_t in_buf, out_buf, res_buf;
    int32_t h, w, cin;    /* input spatial / channels (DENSE/SOFTMAX: h=w=1, cin=K) */
    int32_t oh, ow, cout; /* output                                                    */
    int32_t kh, kw, stride;
    int32_t pad_t, pad_l; /* TF SAME: pad_total//2 on top/left (0 for even input, k=3, s=2) */
    int32_t reserved;   /* CONV_BF16 with w2_off set (round 5, projected shortcut): c2 | stride2 << 12 | h2 << 14 | w2 << 23 -- res_buf is then
                           the BLOCK INPUT [h2, w2, c2], w2_off its 1x1 projection kernel [cout][c2] bf16, shift2_off [scale2 | shift2];
                           PWCONV_F16S / PWCONV_PS / fused kinds: a_log2 (activation pre-scale exponent);
                           DWCONV3X3: 0 = fp32 output, a_log2 > 0 = output stored as split rows scaled by 2^a_log2
                           (act must be ReLU6, c % 32 == 0); 0 otherwise */
    int32_t flags;      /* hsefr_op_flags (round 6; the four bytes were padding before: plans written by older lowerings read as 0) */
    uint64_t w_off;     /* weights; layout depends on kind (see the per-kernel entry points) */
    uint64_t scale_off; /* per-channel scale (DWCONV), descale (PWCONV_F16S)                 */
    uint64_t shift_off; /* per-channel shift / bias                                          */
    uint64_t w2_off;     /* DWPW_F32: pointwise kernel, transposed [cout][cin]; DWPW_F16S: split rows */
    uint64_t shift2_off; /* DWPW_F32: pointwise shift [cout]; DWPW_F16S: [2][cout] descale, shift */
} hsefr_plan_op;

/* ------------------------------------------------------------------------------------ */
/* Engine                                                                                */
/* ------------------------------------------------------------------------------------ */
typedef struct hsefr_engine hsefr_engine;

/* Builds an engine on the CURRENT HIP device: copies the plan's blob to the device and
 * allocates the activation workspace for `max_batch` images.  `plan` is host memory and may
 * be freed after the call.  Replaces tf.import_graph_def + tf.Session (facerec_test.py:41-58). */
int hsefr_engine_create(const void* plan, size_t plan_bytes, int max_batch, hsefr_engine** out);

/* The checks hsefr_engine_create runs on a plan BEFORE it touches the device, alone (no GPU needed): size against the header, buffer ids,
 * blob offsets and operand extents, per-kind shape support, the fusion flags' patterns, the output slots.  HSEFR_OK, or the status and
 * message hsefr_engine_create would return for this blob.  The entry point the sanitizer / fuzz build drives
 * (csrc/build.sh with HSEFR_ASAN=1, tests/test_plan_blob_fuzz_cpu.py). */
int hsefr_plan_validate(const void* plan, size_t plan_bytes);

/* Which kernel runs each op of a plan at batch `n` (no GPU needed): one line per op, "index \t kind \t kernel<template arguments>" -- two
 * kernels joined by " + " where an op is two launches, "(inside op i)" for an op a flagged op's launch covers (hsefr_op_flags).  The table
 * is produced by the launchers themselves running with the launch suppressed: the same shape checks and routing decisions as a forward.
 * `out` receives a NUL-terminated string; HSEFR_ERR_INVALID (with the size needed in the message) if it does not fit. */
int hsefr_plan_describe(const void* plan, size_t plan_bytes, int n, char* out, size_t out_bytes);

/* Batches of at most `max_n` images (default 0 = never) run as ONE hipGraph launch: the op sequence of a (batch size,
 * requested outputs) pair is captured once, reading an engine-owned copy of the input, and replayed.  For callers whose
 * host thread is the bottleneck (the reference calls its session once per image, facerec_test.py:394); on the device
 * itself a batch-1 forward is a chain of ~25 dependent kernels either way (161 us plain, 168 us replayed).
 * hsefr_engine_graph_launches counts forwards served that way. */
int hsefr_engine_set_graph_batch(hsefr_engine* e, int max_n);
long long hsefr_engine_graph_launches(const hsefr_engine* e);

/* Total device bytes the engine holds (weights + workspace). */
size_t hsefr_engine_workspace_bytes(const hsefr_engine* e);
int hsefr_engine_max_batch(const hsefr_engine* e);

/* One pass of the hot path over `n` preprocessed images (d_input: [n,in_h,in_w,in_c] fp32 NHWC,
 * BGR mean-subtracted exactly as facerec_test.py:95-106 leaves it).  Any output pointer may be
 * NULL; a non-NULL pointer for an output the plan does not produce is HSEFR_ERR_INVALID.
 * d_features [n,D] fp32, d_age_probs [n,100] fp32, d_gender [n,1] fp32.  A 16-byte-aligned output pointer is written IN
 * PLACE by the kernel that produces it (no copy; the engine's own buffer of that output is then not written by this
 * forward); any other alignment is served by a device-to-device copy out of the engine's buffer.
 * Replaces tf_sess.run (facerec_test.py:120; facial_analysis.py:109). */
int hsefr_engine_forward(hsefr_engine* e, const void* d_input, int n, void* d_features,
                         void* d_age_probs, void* d_gender, hsefr_stream_t stream);

/* hsefr_engine_forward for callers that hold DECODED, RESIZED images: d_input_u8 is [n,in_h,in_w,3] RGB bytes as the decoder /
 * resizer left them (misc.imresize's output, facerec_test.py:93).  The conversion to float, the channel reversal and the mean
 * subtraction (facerec_test.py:95-106) happen inside the first kernel's window load (csrc/stem4_fused.hip): no fp32 image is ever
 * written.  Only for plans whose first op is the fused stem lowered WITH the mean (lower_graph(..., u8_mean_bgr=...)) on inputs
 * whose edges are multiples of 4: hsefr_engine_accepts_u8 says so, HSEFR_ERR_UNSUPPORTED otherwise.  Results equal
 * hsefr_engine_forward on the preprocessed floats to fp32 round-off (exact products, another summation order). */
int hsefr_engine_accepts_u8(const hsefr_engine* e);
int hsefr_engine_forward_u8(hsefr_engine* e, const void* d_input_u8, int n, void* d_features, void* d_age_probs, void* d_gender,
                            hsefr_stream_t stream);

/* Device pointer of an intermediate activation buffer (contents valid until the next forward); used by the per-layer
 * parity tests.  NULL if `buffer` is out of range.  NOT for the buffers behind the plan's OUTPUT slots: a forward that
 * was given an aligned output pointer writes that tensor into the caller's memory instead, and the engine's buffer keeps
 * whatever an earlier forward left there (run the forward with all three output pointers NULL to have every op write the
 * engine's own buffers -- what Engine.forward_all_layers does). */
void* hsefr_engine_buffer(hsefr_engine* e, int buffer);
/* Asynchronous device-to-device copy of the first `bytes` of an activation buffer into d_dst. */
int hsefr_engine_copy_buffer(hsefr_engine* e, int buffer, void* d_dst, size_t bytes, hsefr_stream_t stream);

/* Per-op device time: with depth > 0 every forward records HIP events around each launch on
 * the forward's own stream into a ring of `depth` event sets (call c uses set c % depth), so a
 * whole timed region can be instrumented without synchronising between steps.  depth 0 = off.
 * op_times_ms waits for set `slot` and writes n_ops elapsed times (ms, 0 for pruned ops). */
int hsefr_engine_set_profiling(hsefr_engine* e, int depth);
long long hsefr_engine_profiled_calls(const hsefr_engine* e);
int hsefr_engine_op_times_ms(hsefr_engine* e, int slot, float* ms, int n_ops);

/* Plans lowered with a declared input bound (HSEFR_OP_STEM3_F16S) check it on the device: *host_flag = 1 if any forward since
 * the last call saw an input value outside the bound (its results are meaningless), else 0; the flag is then cleared.
 * Synchronises `stream`.  Always 0 for plans without a bound. */
int hsefr_engine_input_overflow(hsefr_engine* e, int* host_flag, hsefr_stream_t stream);
/* The same read-and-clear WITHOUT the synchronisation: both operations are only enqueued on `stream`; `pinned_host_flag` must be
 * page-locked host memory (hipHostMalloc / hipHostRegister) that stays valid until the stream has passed this point -- the caller
 * records an event behind the call and reads the int after that event.  For pipelined callers that hand device tensors to
 * hsefr_engine_forward and must not block per batch (TensorFlowInference.extract_batch on a CUDA tensor). */
int hsefr_engine_input_overflow_async(hsefr_engine* e, int* pinned_host_flag, hsefr_stream_t stream);
int hsefr_engine_destroy(hsefr_engine* e); /* replaces tf_sess.close(), facerec_test.py:124-125 */

/* ------------------------------------------------------------------------------------ */
/* Per-kernel entry points (unit parity; the engine calls the same launchers)            */
/* ------------------------------------------------------------------------------------ */

/* Conv2D KxK, Cin=3 -> Cout (multiple of 4, <= 64), stride s, zero padding pad_t/pad_l, + shift + act.
 * x [n,h,w,3], wgt [kh,kw,3,cout] (TF HWIO, BN scale pre-folded as in graph node #30), y [n,oh,ow,cout]. */
int hsefr_conv_c3_bias_act(const float* x, const float* wgt, const float* shift, float* y, int n, int h,
                           int w, int kh, int kw, int stride, int pad_t, int pad_l, int oh, int ow,
                           int cout, int act, hsefr_stream_t stream);

/* DepthwiseConv2dNative 3x3 (+ Mul scale + Add shift + ReLU6; graph nodes #35-39,#44).
 * x [n,h,w,c], wgt [3,3,c] (TF [kh,kw,C,1]), c multiple of 4, stride 1|2, y [n,oh,ow,c]. */
int hsefr_dwconv3x3_bn_relu6(const float* x, const float* wgt, const float* scale, const float* shift,
                             float* y, int n, int h, int w, int c, int stride, int pad_t, int pad_l,
                             int oh, int ow, int act, hsefr_stream_t stream);

/* The same depthwise layer with its result stored PRE-SPLIT for the split-f16 GEMM behind it: y_split holds, per pixel
 * and 32-channel group, one 128-byte "split row" [hi(32 x f16) | lo(32 x f16)] with hi = f16(v * 2^a_log2),
 * lo = f16(v * 2^a_log2 - hi) (same byte size as the fp32 tensor).  act must be HSEFR_ACT_RELU6 (the bound the split
 * needs), a_log2 in 1..12, c multiple of 32. */
int hsefr_dwconv3x3_bn_relu6_split(const float* x, const float* wgt, const float* scale, const float* shift, void* y_split,
                                   int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh, int ow, int act,
                                   int a_log2, hsefr_stream_t stream);

/* Conv2D 1x1 (+ Add shift + ReLU6; graph nodes #45-49) as an fp32-MFMA GEMM:
 * x [m,k] (m = n*h*w pixels, NHWC), wgt_t [cout,k] = the TF kernel [1,1,k,cout] TRANSPOSED,
 * y [m,cout].  k multiple of 32, cout multiple of 64. */
int hsefr_pwconv1x1_bias_relu6(const float* x, const float* wgt_t, const float* shift, float* y,
                               long long m, int k, int cout, int act, hsefr_stream_t stream);

/* The same Conv2D 1x1 + shift + act with every product formed on the f16 MFMA from a two-term f16 split of both
 * operands (3 MFMAs per product, fp32 accumulate; error <= 3*2^-22 per product, i.e. fp32-grade -- see
 * csrc/pwconv_f16s.hip).  x [m,k] fp32 with |x| * 2^a_log2 < 32768 (PRECONDITION: e.g. a ReLU6 producer and a_log2 = 12);
 * w_split = the transposed kernel, per output channel scaled by 2^e_n and split into f16 hi/lo "split rows"
 * [cout][k/32][hi(32) | lo(32)] (byte size = cout*k*4; built by hse_facerec_tf_amd.lowering.split_pointwise_weights);
 * descale[n] = 2^-(e_n + a_log2).  y = act(acc * descale + shift).  k multiple of 32, cout multiple of 64. */
int hsefr_pwconv1x1_f16split(const float* x, const void* w_split, const float* descale, const float* shift, float* y,
                             long long m, int k, int cout, int a_log2, int act, hsefr_stream_t stream);

/* The same layer on PRE-SPLIT activations (x_split = [m][k/32] split rows as hsefr_dwconv3x3_bn_relu6_split writes them,
 * scaled by the 2^a_log2 that descale already undoes): both operands travel global -> LDS by DMA, the K loop carries no
 * VALU (csrc/pwconv_ps.hip).  k multiple of 32, cout multiple of 128; HSEFR_ERR_UNSUPPORTED otherwise.  Same error bound
 * as hsefr_pwconv1x1_f16split (not bit-identical to it: a different MFMA shape sums the products in another order). */
int hsefr_pwconv1x1_presplit(const void* x_split, const void* w_split, const float* descale, const float* shift, float* y,
                             long long m, int k, int cout, int act, hsefr_stream_t stream);

/* hsefr_pwconv1x1_presplit with the global average pool fused into its epilogue: y = [m / map_hw][cout] fp32 means over each map's
 * map_hw pixels (33 <= map_hw <= 288); the pointwise tensor itself is not written. */
int hsefr_pwconv1x1_presplit_gap(const void* x_split, const void* w_split, const float* descale, const float* shift, float* y,
                                 long long m, int k, int cout, int act, int map_hw, hsefr_stream_t stream);

/* hsefr_pwconv1x1_presplit with the next block's depthwise 3x3 / SAME + scale + shift + ReLU6 fused into its epilogue: the result
 * leaves as that depthwise layer's split rows y_split (scaled by 2^out_log2; m / dw_stride^2 pixels).  The m rows are maps of
 * map_hw = map_h * map_w <= 288 pixels (whole maps per GEMM tile); dw_stride 1, or 2 on 12 x 12 / 14 x 14 maps (TF SAME on an even map: no top / left
 * padding).  dw_consts = [11][cout] floats: the depthwise taps (row-major 3x3), then its scale and shift, the last two already
 * multiplied by 2^out_log2. */
int hsefr_pwconv1x1_presplit_dw(const void* x_split, const void* w_split, const float* descale, const float* shift, const float* dw_consts,
                                void* y_split, long long m, int k, int cout, int act, int map_w, int map_hw, int dw_stride, int out_log2,
                                hsefr_stream_t stream);

/* One whole MobileNet block (graph nodes #35-#49 and their later twins) fused, for any c % 32 == 0 and cout % 64 == 0:
 * depthwise 3x3 SAME (stride 1|2) + scale + shift + ReLU6 -> pointwise 1x1 + shift + act with split-f16 products.
 * x [n,h,w,c], wd [3,3,c], w_split / descale as for hsefr_pwconv1x1_f16split (a_log2 in 1..12: the depthwise result is
 * in [0,6]), y [n,oh,ow,cout]. */
int hsefr_dwpw_f16split(const float* x, const float* wd, const float* dscale, const float* dshift, const void* w_split,
                        const float* descale, const float* pshift, float* y, int n, int h, int w, int c, int stride,
                        int pad_t, int pad_l, int oh, int ow, int cout, int a_log2, int act, hsefr_stream_t stream);

/* The stem plus the depthwise half of block 2 (graph nodes #30-#55) in one kernel: ... -> pointwise 32->64 + shift + ReLU6
 * -> depthwise 3x3 stride 2 SAME + scale + shift + act.  wd2 [3,3,64]; y [n,oh2,ow2,64] with h1 = ceil(h/2),
 * oh2 = ceil(h1/2) (same for w); pad_t2/pad_l2 = the stride-2 depthwise's top/left padding (0 for even h1/w1). */
int hsefr_stem2_fused(const float* x, const float* conv_w, const float* conv_shift, const float* wd1, const float* d1scale,
                      const float* d1shift, const void* w_split, const float* descale, const float* pshift, const float* wd2,
                      const float* d2scale, const float* d2shift, float* y, int n, int h, int w, int cpad_t, int cpad_l,
                      int h1, int w1, int pad_t2, int pad_l2, int oh2, int ow2, int a_log2, int act, hsefr_stream_t stream);

/* hsefr_stem2_fused for an input with a declared bound |x| < 2^(15 - in_log2) (csrc/stem3_fused.hip): conv1 on the f16 MFMA
 * from two-term splits.  cw_split / cdescale = the conv kernel [32][27 -> 32] (k = dy*9 + dx*3 + ci) prepared like a
 * pointwise kernel with a_log2 := in_log2 (hse_facerec_tf_amd.lowering.split_pointwise_weights).  d_overflow (device int,
 * may be NULL) is OR-ed with 1 when a value breaks the bound. */
int hsefr_stem3_fused(const float* x, const void* cw_split, const float* cdescale, const float* conv_shift, const float* wd1,
                      const float* d1scale, const float* d1shift, const void* w_split, const float* descale, const float* pshift,
                      const float* wd2, const float* d2scale, const float* d2shift, float* y, int* d_overflow, int n, int h, int w,
                      int cpad_t, int cpad_l, int h1, int w1, int pad_t2, int pad_l2, int oh2, int ow2, int in_log2, int a_log2,
                      int act, hsefr_stream_t stream);

/* The same four layers for inputs whose edges are multiples of 4 (csrc/stem4_fused.hip; hsefr_stem3_fused covers the rest): the
 * input window is converted to f16 once and conv1's MFMA operands are read straight from it (no im2col).  x_is_u8 = 0: x is fp32
 * [n,h,w,3] with the declared bound |x| < 2^(15 - in_log2), conv_shift [32].  x_is_u8 = 1 (in_log2 = 0): x is the RESIZED image as
 * RGB bytes [n,h,w,3] -- the float conversion, channel reversal and mean subtraction of facerec_test.py:95-106 /
 * facial_analysis.py:98-107 are folded into the constants: cw4 is packed channel-reversed and conv_shift is [4][32] =
 * shift - sum over the VALID taps of w * mean, for (pixel in the last conv row ? 2 : 0) + (in the last conv column ? 1 : 0).
 * cw4 / cdescale = the conv kernel in the two-step K layout [2][32][hi 32 | lo 32] f16 (hse_facerec_tf_amd.lowering.stem4_conv_image). */
int hsefr_stem4_fused(const void* x, int x_is_u8, const void* cw4, const float* cdescale, const float* conv_shift, const float* wd1,
                      const float* d1scale, const float* d1shift, const void* w_split, const float* descale, const float* pshift,
                      const float* wd2, const float* d2scale, const float* d2shift, float* y, int* d_overflow, int n, int h, int w,
                      int in_log2, int a_log2, int act, hsefr_stream_t stream);
/* The same four layers, same operands, same arithmetic (the same bits as hsefr_stem4_fused) as a STREAMING kernel (csrc/stem5_stream.hip,
 * round 4): a wave owns 6 output columns of an image and walks down it one output row per step, every depthwise output accumulated in
 * registers as its input rows arrive -- no vertical halo, no barriers, twelve independent waves per CU.  What the engine runs for
 * HSEFR_OP_STEM3_F16S on inputs whose edges are multiples of 4. */
int hsefr_stem5_stream(const void* x, int x_is_u8, const void* cw4, const float* cdescale, const float* conv_shift, const float* wd1,
                      const float* d1scale, const float* d1shift, const void* w_split, const float* descale, const float* pshift,
                      const float* wd2, const float* d2scale, const float* d2shift, float* y, int* d_overflow, int n, int h, int w,
                      int in_log2, int a_log2, int act, hsefr_stream_t stream);

/* One whole early MobileNet block (graph nodes #35-#49) fused: depthwise 3x3 SAME (stride 1|2) + scale + shift + ReLU6
 * -> pointwise 1x1 + shift + ReLU6.  x [n,h,w,c], wd [3,3,c], wp_t [cout,c] (TF kernel transposed), y [n,oh,ow,cout];
 * c in {32,64}, cout in {64,128}; HSEFR_ERR_UNSUPPORTED otherwise (callers fall back to the two separate kernels). */
int hsefr_dwpw_fused(const float* x, const float* wd, const float* dscale, const float* dshift, const float* wp_t,
                     const float* pshift, float* y, int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh,
                     int ow, int cout, hsefr_stream_t stream);

/* Mean over H,W (graph node #230): x [n,hw,c] -> y [n,c]; c multiple of 4. */
int hsefr_gap(const float* x, float* y, int n, int hw, int c, hsefr_stream_t stream);

/* MatMul + BiasAdd (+ Relu | Sigmoid) (graph nodes #232-234, #236-238): x [n,k], wgt [k,cout]
 * (TF layout), y [n,cout]. */
int hsefr_dense(const float* x, const float* wgt, const float* bias, float* y, int n, int k, int cout,
                int act, hsefr_stream_t stream);

/* Softmax over the last axis (graph node #241): x,y [n,c], c <= 1024. */
int hsefr_softmax(const float* x, float* y, int n, int c, hsefr_stream_t stream);

/* The age / gender heads in ONE launch (graph nodes #232-241; sess.run([age, gender, feats]) at facial_analysis.py:109):
 *   hidden = relu(x . w1 + b1)                      x [n,k], w1 [k,256], hidden [n,256]
 *   age_probs = softmax(hidden . wa + ba)           wa [256,a], a <= 128; logits [n,a], age_probs [n,a]
 *   gender = sigmoid(hidden . wg + bg)              wg [256,1]; gender [n,1]
 * A lane owns four adjacent columns (16-byte weight loads), the workgroup's sixteen waves split k: fixed summation order (bit-identical
 * run to run, independent of the batch), fp32 grade, round-off apart from hsefr_dense's four slices.
 * k % 256 == 0, k <= 2048.  What the engine runs for a DENSE op flagged HSEFR_OPF_HEADS. */
int hsefr_heads_fused(const float* x, const float* w1, const float* b1, const float* wa, const float* ba, const float* wg, const float* bg,
                      float* hidden, float* logits, float* age_probs, float* gender, int n, int k, int a, hsefr_stream_t stream);

/* ---- bf16 ResNet-50 path (vgg2_resnet.pb, facerec_test.py:213): activations bf16 NHWC, fp32 accumulate ---- */

/* KxK convolution (1x1 / 3x3, stride 1|2, explicit zero padding) as a bf16-MFMA implicit GEMM with the folded
 * BatchNorm in the epilogue: y = act(scale[n]*conv + shift[n] (+ res)).  x [n,h,w,c] bf16, wgt_t [cout][kh*kw*c]
 * bf16 (k = (kh*KW + kw)*c + ci), res/y [n,oh,ow,cout] bf16 (res may be NULL); c and cout multiples of 64. */
int hsefr_conv_bf16(const void* x, const void* wgt_t, const float* scale, const float* shift, const void* res, void* y,
                    int n, int h, int w, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t,
                    int pad_l, int act, hsefr_stream_t stream);

/* The "increase" 1x1 convolution of a ResNet stage's first bottleneck with its PROJECTED shortcut in the same launch (round 5):
 *   y[p, :] = act( bf16( scale * (x[p, :] . wgt_t) + shift ) + bf16( scale2 * (x2[n, oh * stride2, ow * stride2, :] . wgt2_t) + shift2 ) )
 * -- the rounding points of the two-launch form (hsefr_conv_bf16 for the projection, then hsefr_conv_bf16 with res): the projection's
 * [n, oh, ow, cout] tensor is never written.  x [n,oh,ow,c], wgt_t [cout][c], x2 [n,h2,w2,c2] (the block's input), wgt2_t [cout][c2],
 * y [n,oh,ow,cout], all bf16; c, c2, cout multiples of 64.  What the engine runs for an HSEFR_OP_CONV_BF16 whose w2_off is set
 * (res_buf = the block input, shift2 = [scale2 | shift2], reserved = c2 | stride2 << 12 | h2 << 14 | w2 << 23). */
int hsefr_conv1x1_proj_bf16(const void* x, const void* wgt_t, const float* scale, const float* shift, const void* x2, const void* wgt2_t,
                            const float* scale2, const float* shift2, void* y, int n, int oh, int ow, int c, int cout, int c2, int stride2,
                            int h2, int w2, int act, hsefr_stream_t stream);

/* A 1x1 convolution whose RESIDUAL is a stride view of a larger map (round 6):
 *   y[(i, oy, ox), :] = act( bf16( scale * (x[(i, oy, ox), :] . wgt_t) + shift ) + res[i, oy * res_stride, ox * res_stride, :] )
 * x [n,oh,ow,c], wgt_t [cout][c], res [n,h2,w2,cout], y [n,oh,ow,cout], all bf16; c, cout multiples of 64; (oh - 1) * res_stride < h2.  The last
 * bottleneck of a ResNet stage feeds only stride-2 1x1 layers: its 3x3 and increase layers run at the pixels those layers read
 * (lowering.subsample_stage_tails; results identical at those pixels), and the block's shortcut is still the full-size map of the block
 * before it.  What the engine runs for an HSEFR_OP_CONV_BF16 with a residual, no w2_off and reserved = res_stride << 12 | h2 << 14 | w2 << 23. */
int hsefr_conv1x1_sres_bf16(const void* x, const void* wgt_t, const float* scale, const float* shift, const void* res, void* y, int n, int oh,
                            int ow, int c, int cout, int res_stride, int h2, int w2, int act, hsefr_stream_t stream);

/* Two chained 1x1 convolutions at the same pixels in ONE launch (round 6; csrc/conv1x1_pair_bf16.hip): a bottleneck's increase layer
 * and the next bottleneck's reduce layer,
 *   y1[p, :] = act1( bf16( scale1 * (x[p, :] . w1_t) + shift1 ) + R[p, :] ),   R = res[p, :]  or  bf16( scale_p * (x2[p, :] . wp_t) + shift_p )
 *   y2[p, :] = act2( bf16( scale2 * (y1[p, :] . w2_t) + shift2 ) )
 * -- the rounding points of hsefr_conv_bf16 (+ res) / hsefr_conv1x1_proj_bf16 followed by hsefr_conv_bf16; y1 is stored and ALSO kept in
 * registers as the second product's operand, so it is not read back.  x [pixels,c], w1_t [cout1][c], res / y1 [pixels,cout1], w2_t
 * [cout2][cout1], y2 [pixels,cout2], all bf16; exactly one of res and (x2 [pixels,c2], wp_t [cout1][c2], scale_p, shift_p) is given
 * (c2 = 0 without a projection).  Covered: c = 64, cout1 = 256, cout2 = 64, c2 in {0, 64} (ResNet-50's 56-pixel stage).  What the
 * engine runs for a CONV_BF16 op flagged HSEFR_OPF_PAIR_NEXT. */
int hsefr_conv1x1_pair_bf16(const void* x, const void* w1_t, const float* scale1, const float* shift1, const void* res, const void* x2,
                            const void* wp_t, const float* scale_p, const float* shift_p, void* y1, const void* w2_t, const float* scale2,
                            const float* shift2, void* y2, long long pixels, int c, int cout1, int cout2, int c2, int act1, int act2,
                            hsefr_stream_t stream);

/* The same launch with y1 STORED at the pixels with even row and column only (a compact [n,(h+1)/2,(w+1)/2,cout1] map; y2 unchanged, from
 * every pixel): for a y1 whose only other reader takes every second pixel of it.  x [n,h,w,c].  What the engine runs for a PAIR_NEXT op that
 * also carries HSEFR_OPF_OUT_SUB2. */
int hsefr_conv1x1_pair_sub2_bf16(const void* x, const void* w1_t, const float* scale1, const float* shift1, const void* res, const void* x2,
                                 const void* wp_t, const float* scale_p, const float* shift_p, void* y1, const void* w2_t, const float* scale2,
                                 const float* shift2, void* y2, int n, int h, int w, int c, int cout1, int cout2, int c2, int act1, int act2,
                                 hsefr_stream_t stream);

/* ResNet stem: 7x7 / stride 2 / pad 3 conv over the fp32 image [n,h,w,3] -> [n,oh,ow,64] bf16, + scale + shift + act.
 * wgt_t [64][256] bf16 with k = dy*32 + dx*3 + ci, zero padded. */
int hsefr_stem7x7_bf16(const float* x, const void* wgt_t, const float* scale, const float* shift, void* y, int n, int h,
                       int w, int oh, int ow, int act, hsefr_stream_t stream);

/* 3x3 / stride 2 max-pool, bf16 NHWC; windows are clipped to the image (Caffe ceil mode = pad 0, TF SAME = its pads). */
int hsefr_maxpool3x3s2_bf16(const void* x, void* y, int n, int h, int w, int c, int oh, int ow, int pad_t, int pad_l,
                            hsefr_stream_t stream);

/* Mean over H,W of bf16 activations -> fp32 [n,c] (pool5/7x7_s1). */
int hsefr_gap_bf16(const void* x, float* y, int n, int hw, int c, hsefr_stream_t stream);

/* ---- device-side image preparation (facerec_test.py:80-112, facial_analysis.py:95-107), bit-exact ---------------- */

/* colour/mean handling after the resize:
 * 0 = BGR - mean3 evaluated in float64 then cast (facerec_test.py:95-106 feeding a float32 placeholder),
 * 1 = RGB, x/127.5 - 1 (facerec_test.py:108-110), 2 = BGR - mean3 in float32 (facial_analysis.py:101-107),
 * 3 = none: d_out receives the resized RGB BYTES [n,oh,ow,3] (uint8, a quarter of the fp32 tensor) for
 * hsefr_engine_forward_u8, whose first kernel folds conversion, channel reversal and mean into its window load. */
typedef enum hsefr_color_mode { HSEFR_COLOR_BGR_MEAN_F64 = 0, HSEFR_COLOR_RGB_UNIT = 1, HSEFR_COLOR_BGR_MEAN_F32 = 2,
                                HSEFR_COLOR_NONE_U8 = 3 } hsefr_color_mode;

/* misc.imresize(img, (oh, ow), 'bilinear') (= PIL BILINEAR, antialiased, 8-bit fixed point) + colour handling.
 * d_in [n,H,W,3] u8 RGB, d_tmp [n,H,ow,3] u8 scratch, d_out [n,oh,ow,3] f32.  The per-axis coefficient tables are
 * Pillow's precompute_coeffs/normalize_coeffs_8bpc output (device int32 arrays): first tap, tap count and
 * `ksize` 22-bit weights per output coordinate.  mean3 is HOST memory (3 doubles, BGR order). */
int hsefr_preprocess_pil_u8(const unsigned char* d_in, unsigned char* d_tmp, float* d_out, int n, int H, int W, int oh, int ow,
                            const int* d_xmin, const int* d_xcnt, const int* d_xcoef, int xksize, const int* d_ymin,
                            const int* d_ycnt, const int* d_ycoef, int yksize, int color_mode, const double* mean3,
                            hsefr_stream_t stream);

/* cv2.resize(img, (ow, oh)) INTER_LINEAR on 8-bit images + colour handling.  Per-axis tap tables (device int32):
 * the two source indices and the 11-bit weight of the second tap for every output coordinate. */
int hsefr_preprocess_cv_u8(const unsigned char* d_in, float* d_out, int n, int H, int W, int oh, int ow, const int* d_x0,
                           const int* d_x1, const int* d_wx1, const int* d_y0, const int* d_y1, const int* d_wy1,
                           int color_mode, const double* mean3, hsefr_stream_t stream);

/* preprocessing.normalize(X, 'l2') (facerec_test.py:401): rows of x [n,d] scaled in place-free
 * fashion into y; zero rows stay zero (sklearn divides by 1 then). */
int hsefr_l2_normalize(const float* x, float* y, int n, int d, hsefr_stream_t stream);

/* KNeighborsClassifier(n_neighbors=1, p=2).kneighbors (facerec_test.py:422,200-207): for every
 * query row q [nq,d] the index (int32) and squared L2 distance of its nearest gallery row g [ng,d];
 * ties resolve to the lowest gallery index.  d multiple of 4. */
int hsefr_nn1(const float* q, const float* g, int nq, int ng, int d, int* nn_index, float* nn_dist2,
              hsefr_stream_t stream);
/* Large searches run on the split-f16 GEMM and need a stream-ordered workspace; when even its smallest form cannot be allocated the
 * search runs on the workspace-free kernel instead (same nearest neighbours up to last-bit ties, far slower at 10^5 x 10^5).  This
 * counts those searches since the library was loaded, so a perf cliff can be traced to the allocator (hsefr_last_error_string says which). */
long long hsefr_nn1_fallbacks(void);

/* ---- generic small-CNN kernels: the MTCNN detection cascade (facial_analysis.py:334-352,478-604; mtcnn.pb) ---------- */

/* Conv2D (any KxK, stride, explicit top/left zero padding, any channel counts) + BiasAdd + optional PReLU, NHWC fp32.
 * x [n,h,w,c], wgt [kh,kw,c,cout] (TF HWIO), bias [cout] or NULL, alpha [cout] or NULL (PReLU slope), y [n,oh,ow,cout].
 * A fully-connected layer is the VALID convolution whose kernel covers the whole map (MatMul weights reshaped). */
int hsefr_conv2d_direct(const float* x, const float* wgt, const float* bias, const float* alpha, float* y, int n, int h, int w, int c,
                        int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, hsefr_stream_t stream);

/* cv2.resize(frame, (dw, dh), interpolation=cv2.INTER_AREA) of a uint8 RGB frame [sh,sw,3] for one level of the MTCNN image
 * pyramid (facial_analysis.py:507), with the cascade's normalisation (v - 127.5) * 0.0078125 and its (W, H) transposition
 * fused in: d_dst float32 [dw, dh, 3].  OpenCV's 8-bit semantics (float32 box accumulation, rounding to uint8). */
int hsefr_mtcnn_pyramid_level(const unsigned char* d_frame, float* d_dst, int sh, int sw, int dh, int dw, hsefr_stream_t stream);

/* MTCNN's box logic on the device (generateBoundingBox / nms / bbreg / rerec / pad, facial_analysis.py:354-476, as driven by
 * mtcnn_detect_faces :478-604): float64 boxes, float32 scores / regressions / landmarks, NumPy's rounding points; greedy NMS by
 * descending score, ties by ascending index.  `counters` = int32[8] on the device, zeroed by the caller per frame:
 * [0] survivors of the pyramid levels so far, [1] / [2] / [3] boxes after stage 1 / 2 / 3, [4] overflow (a list exceeded
 * hsefr_mtcnn_post_capacity() boxes: the frame must be redone on the host).  All lists hold hsefr_mtcnn_post_capacity() rows.
 *   stage1_level : P-Net maps of one level, prob [w,h,2] and reg [w,h,4] (the nets see the transposed frame), cells with
 *                  prob >= thr -> boxes at 1/scale -> NMS 0.5 -> appended to found [cap,9] = x1,y1,x2,y2,score,reg0..3
 *   stage1_finish: NMS 0.7 over found -> regression -> square -> truncate -> boxes [cap,5] and the crop table [cap,8] of
 *                  hsefr_mtcnn_crops
 *   stage_finish : stage 2 (R-Net: prob [n,2], reg [n,4]; score > thr -> NMS 0.7 -> regression -> square -> truncate -> boxes + crop
 *                  table) or stage 3 (O-Net: + pts [n,10]; score > thr -> landmarks -> regression -> NMS 0.7 'Min' -> boxes [cap,5]
 *                  and points [cap,10] float32)
 *   nms          : plain NMS of boxes [n,5] (n <= capacity): keep[] = kept indices in pick order                                  */
int hsefr_mtcnn_post_capacity(void);
int hsefr_mtcnn_stage1_level(const float* prob, const float* reg, int w, int h, double scale, float thr, double* found, int* counters,
                             hsefr_stream_t stream);
int hsefr_mtcnn_stage1_finish(const double* found, int* counters, double* boxes, int* crop_table, int img_w, int img_h,
                              hsefr_stream_t stream);
int hsefr_mtcnn_stage_finish(int stage, const double* boxes_in, int n, const float* prob, const float* reg, const float* pts, float thr,
                             double* boxes_out, int* crop_table, float* points_out, int* counters, int img_w, int img_h,
                             hsefr_stream_t stream);
int hsefr_mtcnn_nms(const double* boxes, int n, double thr, int use_min, int* keep, int* n_keep, hsefr_stream_t stream);

/* The R-Net / O-Net inputs of the cascade (facial_analysis.py:546,575): for each of n boxes, the box-sized tile of the frame
 * (zero outside it) resized to size x size with INTER_AREA in float64, normalised and transposed: d_dst float32
 * [n, size, size, 3].  d_boxes int32 [n][8] = {x1, y1, x2, y2 (1-based inclusive window clipped to the frame), tx1, ty1
 * (where the window lands in the tile), bw, bh (tile size)} as the reference's pad() computes them. */
int hsefr_mtcnn_crops(const unsigned char* d_frame, const int* d_boxes, float* d_dst, int sh, int sw, int n, int size,
                      hsefr_stream_t stream);

/* hsefr_conv2d_f32 as an implicit GEMM on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, 155 TFLOP/s):
 * same arguments and results to fp32 round-off (another summation order); cout multiple of 64.  What HSEFR_OP_CONV_F32 runs
 * when it covers the layer (csrc/conv_f32_mfma.hip): the fp32-grade mode of vgg2_resnet.pb (facerec_test.py:213). */
int hsefr_conv2d_f32_mfma(const float* x, const float* wgt, const float* scale, const float* shift, const float* res, float* y, int n,
                          int h, int w, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, int act,
                          hsefr_stream_t stream);

/* General Conv2D in exact fp32 (FMA chains) + per-channel scale (NULL = 1) + shift (NULL = 0) + optional residual + act:
 * x [n,h,w,c], wgt [kh,kw,c,cout] (TF HWIO), res / y [n,oh,ow,cout], cout multiple of 4.  The fp32-grade mode of the
 * ResNet-style graphs; an order of magnitude slower than hsefr_conv_bf16. */
int hsefr_conv2d_f32(const float* x, const float* wgt, const float* scale, const float* shift, const float* res, float* y, int n, int h,
                     int w, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, int act,
                     hsefr_stream_t stream);

/* conv1 7x7/2 pad 3 (3 -> 64) + scale + shift + ReLU + max-pool 3x3/2 in one kernel: x fp32 [n,h,w,3], wgt_t as
 * hsefr_stem7x7_bf16, y bf16 [n,ph,pw,64].  Pool windows start at 2 * p - pool_pad and are clipped to the conv map
 * (pool_pad 0 with ph = ceil((oh - 3) / 2) + 1 is Caffe's ceil mode; TF SAME / an explicit Pad pass 0 or 1).  Runs the streaming kernel
 * (csrc/stem7s_stream.hip) where the launch fits 32-bit byte offsets, the patch kernel (csrc/stem7x7_pool.hip) otherwise: same rounding
 * points, fp32 accumulation order differs (rare one-ulp bf16 differences between the two). */
int hsefr_stem7x7_pool_bf16(const float* x, const void* wgt_t, const float* scale, const float* shift, void* y, int n, int h, int w,
                            int ph, int pw, int pool_pad_t, int pool_pad_l, hsefr_stream_t stream);

/* MaxPool k x k / stride with windows clipped to the image (TF SAME/VALID: pass the TF pads), NHWC fp32. */
int hsefr_maxpool_f32(const float* x, float* y, int n, int h, int w, int c, int oh, int ow, int k, int stride, int pad_t, int pad_l,
                      hsefr_stream_t stream);

/* sklearn.metrics.pairwise_distances(X[, Y]) (euclidean; facial_clustering_test.py:396, and the feature term of
 * process_photos.py:46-51): out[i,j] = |x_i - y_j|, x [n,d], y [m,d], out [n,m]; d multiple of 8.  Passing the same
 * pointer for x and y gives an exactly zero diagonal.  Computed as |x|^2+|y|^2-2x.y on the fp32 MFMA: absolute error
 * ~1e-7*(|x|^2+|y|^2) on the SQUARED distance, i.e. distances below ~1e-2 (near-duplicates) carry up to 6e-4. */
int hsefr_pairwise_dist(const float* x, const float* y, int n, int m, int d, float* out, hsefr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HSEFR_H */
