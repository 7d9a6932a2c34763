// libhsefr C ABI: plan loader + static-graph executor (include/hsefr.h).
//
// The engine is the replacement for tf.import_graph_def + tf.Session + Session.run
// (facerec_test.py:41-58,117-120; facial_analysis.py:55-58,109): a frozen graph that the host
// side has lowered to a linear op list ("plan") runs as a fixed sequence of kernel launches on
// ONE stream over pre-allocated NHWC activation buffers.  No allocation, no synchronisation and
// no host<->device copy happens inside hsefr_engine_forward, so a forward can be captured into
// a hipGraph by the caller.
#include <string.h>

#include <cxxabi.h>
#include <dlfcn.h>
#include <elf.h>

#include <map>
#include <new>
#include <string>
#include <vector>

#include "common.h"

namespace hsefr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace hsefr

using namespace hsefr;

static_assert(sizeof(hsefr_plan_header) == 64, "plan header layout is part of the ABI");
static_assert(sizeof(hsefr_plan_buffer) == 16, "plan buffer layout is part of the ABI");
static_assert(sizeof(hsefr_plan_op) == 112, "plan op layout is part of the ABI");

struct hsefr_engine {
    hsefr_plan_header hdr;
    std::vector<hsefr_plan_buffer> bufs;
    std::vector<hsefr_plan_op> ops;
    std::vector<void*> d_bufs;
    char* d_blob = nullptr;
    int* d_overflow = nullptr;        // input-bound flag of STEM3 ops (hsefr_engine_input_overflow)
    std::vector<void*> d_derived;     // per op: an operand the engine derives from the blob once (STEM7X7_POOL_BF16: the weight image in the
                                      // streaming kernel's fragment order), or null
    size_t device_bytes = 0;
    int max_batch = 0;
    int device = 0;
    // profiling: a ring of `depth` event sets (n_ops + 1 events each), one set per forward
    // call, so a whole timed region can be instrumented without a sync between steps
    int prof_depth = 0;
    long long prof_calls = 0;
    std::vector<hipEvent_t> events;  // depth * (n_ops + 1)
    // Optional: for n <= graph_max_n the op sequence of a (n, requested outputs) pair is captured ONCE into a hipGraph --
    // reading an engine-owned copy of the input, so the captured pointers never change -- and replayed.  OFF by default:
    // measured on MI355X / ROCm 7.2 a batch-1 forward takes 161 us as 25 plain launches and 168 us as one graph launch
    // (the time is the chain of 25 dependent few-microsecond kernels on the device, not their submission).
    int graph_max_n = 0;
    void* d_in_stage = nullptr;
    hipStream_t cap_stream = nullptr;
    struct GraphEntry { int n; int mask; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    long long graph_launches = 0;
};

// stem4_fused.hip hard-codes TensorFlow's SAME geometry for inputs whose edges are multiples of 4 (conv1 and the stride-2
// depthwise pad at the bottom / right only, H1 = h / 2, OH2 = h / 4) and its uint8 mean-folded shifts assume it too: an op with
// any other padding (VALID, a hand-made plan) keeps stem3_fused.hip, which honours pad_t / pad_l / pad3 / oh / ow.
static bool stem4_route(const hsefr_plan_op& o) {
    return stem4_fused_supported(o.cin, 32, o.cout, o.stride, 1, 2, o.kh, o.kw & 15, o.h, o.w) && o.pad_t == 0 && o.pad_l == 0 &&
           ((o.kw >> 4) & 3) == 0 && o.oh == o.h / 4 && o.ow == o.w / 4;
}

static const void* blob_ptr(const hsefr_engine* e, uint64_t off) {
    return off == HSEFR_NO_OFFSET ? nullptr : (const void*)(e->d_blob + off);
}

// Buffer table of ONE forward: the engine's activation buffers, with the buffers of the requested outputs replaced by
// the caller's pointers -- the producing kernel writes its result where the caller wants it (and any later op that
// reads that tensor, e.g. the dense heads behind the pooled features, reads it from there): no copy-out.
static void* buf_ptr(const std::vector<void*>& tab, int id, const void* d_input) {
    if (id == HSEFR_BUF_INPUT) return const_cast<void*>(d_input);
    if (id < 0 || id >= (int)tab.size()) return nullptr;
    return tab[id];
}

static int validate_plan(const hsefr_plan_header& h, const hsefr_plan_buffer* bufs, const hsefr_plan_op* ops) {
    for (uint32_t i = 0; i < h.n_ops; ++i) {
        const hsefr_plan_op& o = ops[i];
        auto okbuf = [&](int b, bool allow_input) {
            return (b >= 0 && b < (int)h.n_buffers) || (allow_input && b == HSEFR_BUF_INPUT);
        };
        HSEFR_REQUIRE(okbuf(o.in_buf, true) && okbuf(o.out_buf, false), HSEFR_ERR_INVALID,
                      "plan op %u: bad buffer ids (%d -> %d)", i, o.in_buf, o.out_buf);
        HSEFR_REQUIRE(o.res_buf == HSEFR_BUF_NONE || okbuf(o.res_buf, false), HSEFR_ERR_INVALID,
                      "plan op %u: bad residual buffer %d", i, o.res_buf);
        // every dimension in a range the launch wrappers' 32-bit arithmetic is safe in (found by the sanitizer fuzz of round 6: an op with
        // oh = INT_MAX and ow = 0 passed the byte checks below and overflowed a tile count), and the last output's first tap inside the
        // image -- a kernel treats whatever lies beyond the image as padding, never as memory to read
        HSEFR_REQUIRE(o.h >= 1 && o.h <= 32768 && o.w >= 1 && o.w <= 32768 && o.oh >= 1 && o.oh <= 32768 && o.ow >= 1 && o.ow <= 32768 && o.cin >= 1 &&
                          o.cin <= 65536 && o.cout >= 1 && o.cout <= 65536 && o.kh >= 1 && o.kh <= 15 && o.kw >= 1 && o.kw <= 63 && o.stride >= 1 &&
                          o.stride <= 8 && o.pad_t >= 0 && o.pad_t <= 15 && o.pad_l >= 0 && o.pad_l <= 15,
                      HSEFR_ERR_INVALID, "plan op %u: a dimension is out of range (%dx%dx%d -> %dx%dx%d, kernel %dx%d / %d, pad %d %d)", i, o.h, o.w, o.cin,
                      o.oh, o.ow, o.cout, o.kh, o.kw, o.stride, o.pad_t, o.pad_l);
        {
            const bool two_stages = o.kind == HSEFR_OP_STEM2_F16S || o.kind == HSEFR_OP_STEM3_F16S || o.kind == HSEFR_OP_STEM7X7_POOL_BF16;
            const long long st = two_stages ? 4 : o.stride, slack = two_stages ? 3 : 0;
            const bool spatial = !(o.kind == HSEFR_OP_DENSE || o.kind == HSEFR_OP_SOFTMAX || o.kind == HSEFR_OP_GAP || o.kind == HSEFR_OP_GAP_BF16 ||
                                   o.kind == HSEFR_OP_PWCONV_PS_GAP);
            HSEFR_REQUIRE(!spatial || ((long long)(o.oh - 1) * st - o.pad_t - slack < o.h && (long long)(o.ow - 1) * st - o.pad_l - slack < o.w),
                          HSEFR_ERR_INVALID, "plan op %u: a %dx%d output does not fit a %dx%d input at stride %lld", i, o.oh, o.ow, o.h, o.w, st);
        }
        for (uint64_t off : {o.w_off, o.scale_off, o.shift_off, o.w2_off, o.shift2_off})
            HSEFR_REQUIRE(off == HSEFR_NO_OFFSET || (off < h.blob_bytes && off % 16 == 0), HSEFR_ERR_INVALID,
                          "plan op %u: blob offset %llu out of range / unaligned", i, (unsigned long long)off);
        const bool out_bf16 = o.kind == HSEFR_OP_CONV_BF16 || o.kind == HSEFR_OP_MAXPOOL_BF16 || o.kind == HSEFR_OP_STEM7X7_BF16 ||
                              o.kind == HSEFR_OP_STEM7X7_POOL_BF16;
        const bool in_bf16 = o.kind == HSEFR_OP_CONV_BF16 || o.kind == HSEFR_OP_MAXPOOL_BF16 || o.kind == HSEFR_OP_GAP_BF16;
        const uint64_t out_bytes = (uint64_t)o.oh * o.ow * o.cout * (out_bf16 ? 2 : 4);
        const uint64_t out_cap = bufs[o.out_buf].elems_per_image * bufs[o.out_buf].elem_bytes;
        HSEFR_REQUIRE(out_bytes <= out_cap, HSEFR_ERR_INVALID, "plan op %u: output %llu bytes/image exceeds buffer %d (%llu)", i,
                      (unsigned long long)out_bytes, o.out_buf, (unsigned long long)out_cap);
        const bool from_registers = i > 0 && (ops[i - 1].flags & HSEFR_OPF_OUT_SUB2) && o.in_buf == ops[i - 1].out_buf;      // (checked with the flag below)
        if (o.in_buf >= 0 && !from_registers) {
            const uint64_t in_bytes = (uint64_t)o.h * o.w * o.cin * (in_bf16 ? 2 : 4);
            HSEFR_REQUIRE(in_bytes <= bufs[o.in_buf].elems_per_image * bufs[o.in_buf].elem_bytes, HSEFR_ERR_INVALID,
                          "plan op %u: input exceeds buffer %d", i, o.in_buf);
        }
        HSEFR_REQUIRE(o.res_buf != o.out_buf, HSEFR_ERR_INVALID, "plan op %u: residual aliases the output", i);
        // every operand a kind reads must be present and fit the blob with its whole extent (a truncated or hand-made
        // plan is HSEFR_ERR_INVALID, never an out-of-bounds device read)
        auto need = [&](uint64_t off, uint64_t bytes, const char* what) {
            if (off != HSEFR_NO_OFFSET && off + bytes <= h.blob_bytes) return true;
            set_error("plan op %u (kind %u): operand %s missing or %llu bytes beyond the %llu-byte blob", i, o.kind, what,
                      (unsigned long long)bytes, (unsigned long long)h.blob_bytes);
            return false;
        };
        const uint64_t co = o.cout, ci = o.cin, kk = (uint64_t)o.kh * o.kw;
        switch (o.kind) {
            case HSEFR_OP_GAP: case HSEFR_OP_SOFTMAX: case HSEFR_OP_MAXPOOL_BF16: case HSEFR_OP_GAP_BF16: case HSEFR_OP_MAXPOOL_F32:
                break;
            case HSEFR_OP_CONV_F32:
                if (!need(o.w_off, kk * ci * co * 4, "kernel")) return HSEFR_ERR_INVALID;
                if (o.scale_off != HSEFR_NO_OFFSET && !need(o.scale_off, co * 4, "scale")) return HSEFR_ERR_INVALID;
                if (o.shift_off != HSEFR_NO_OFFSET && !need(o.shift_off, co * 4, "shift")) return HSEFR_ERR_INVALID;
                HSEFR_REQUIRE(o.cout % 4 == 0, HSEFR_ERR_UNSUPPORTED, "plan op %u: fp32 convolution with cout=%d (must be a multiple of 4)", i, o.cout);
                if (o.reserved != 0) {      // strided residual, as HSEFR_OP_CONV_BF16's (round 6): only the fp32-MFMA kernel reads one
                    const int k2 = o.reserved & 0xFFF, st2 = (o.reserved >> 12) & 3, h2 = (o.reserved >> 14) & 0x1FF, w2 = (o.reserved >> 23) & 0x1FF;
                    HSEFR_REQUIRE(k2 == 0 && o.kh == 1 && o.kw == 1 && o.stride == 1 && o.pad_t == 0 && o.pad_l == 0 && o.res_buf >= 0 && st2 >= 1 &&
                                      o.oh == o.h && o.ow == o.w && o.oh * o.ow > 1 && o.ow > 1 && (o.oh - 1) * st2 < h2 && (o.ow - 1) * st2 < w2 &&
                                      conv_f32_mfma_supported(o.cin, o.cout),
                                  HSEFR_ERR_INVALID, "plan op %u: bad strided residual (stride %d, %dx%d)", i, st2, h2, w2);
                    HSEFR_REQUIRE((uint64_t)h2 * w2 * co * 4 <= bufs[o.res_buf].elems_per_image * bufs[o.res_buf].elem_bytes, HSEFR_ERR_INVALID,
                                  "plan op %u: strided residual exceeds buffer %d", i, o.res_buf);
                }
                break;
            case HSEFR_OP_CONV_C3:
                if (!need(o.w_off, kk * ci * co * 4, "kernel") || !need(o.shift_off, co * 4, "shift")) return HSEFR_ERR_INVALID;
                break;
            case HSEFR_OP_DWCONV3X3:
                if (!need(o.w_off, 9 * ci * 4, "kernel") || !need(o.scale_off, ci * 4, "scale") || !need(o.shift_off, ci * 4, "shift"))
                    return HSEFR_ERR_INVALID;
                HSEFR_REQUIRE(o.reserved == 0 || (o.reserved > 0 && o.reserved <= 12 && o.act == HSEFR_ACT_RELU6 && o.cin % 32 == 0),
                              HSEFR_ERR_INVALID, "plan op %u: split-row depthwise output needs ReLU6, c %% 32 == 0 and a_log2 in [1, 12]", i);
                break;
            case HSEFR_OP_PWCONV_F32:
                if (!need(o.w_off, ci * co * 4, "kernel") || !need(o.shift_off, co * 4, "shift")) return HSEFR_ERR_INVALID;
                break;
            case HSEFR_OP_DENSE:
                if (!need(o.w_off, ci * co * 4, "kernel")) return HSEFR_ERR_INVALID;
                if (o.shift_off != HSEFR_NO_OFFSET && !need(o.shift_off, co * 4, "bias")) return HSEFR_ERR_INVALID;
                break;
            case HSEFR_OP_CONV_BF16:
                if (!need(o.w_off, kk * ci * co * 2, "kernel") || !need(o.scale_off, co * 4, "scale") || !need(o.shift_off, co * 4, "shift"))
                    return HSEFR_ERR_INVALID;
                if (o.w2_off != HSEFR_NO_OFFSET) {
                    // projected shortcut (round 5): res_buf is the BLOCK INPUT [h2, w2, k2], w2 its 1x1 kernel [cout][k2], shift2 =
                    // [scale2 | shift2]; reserved = k2 | stride << 12 | h2 << 14 | w2 << 23
                    const int k2 = o.reserved & 0xFFF, st2 = (o.reserved >> 12) & 3, h2 = (o.reserved >> 14) & 0x1FF, w2 = (o.reserved >> 23) & 0x1FF;
                    HSEFR_REQUIRE(o.kh == 1 && o.kw == 1 && o.stride == 1 && o.res_buf >= 0 && k2 > 0 && k2 % 64 == 0 && st2 >= 1 &&
                                      (o.oh - 1) * st2 < h2 && (o.ow - 1) * st2 < w2,
                                  HSEFR_ERR_INVALID, "plan op %u: bad projected shortcut (k2 %d, stride %d, %dx%d)", i, k2, st2, h2, w2);
                    if (!need(o.w2_off, (uint64_t)k2 * co * 2, "projection kernel") || !need(o.shift2_off, 2 * co * 4, "projection scale / shift"))
                        return HSEFR_ERR_INVALID;
                    HSEFR_REQUIRE((uint64_t)h2 * w2 * k2 * 2 <= bufs[o.res_buf].elems_per_image * bufs[o.res_buf].elem_bytes, HSEFR_ERR_INVALID,
                                  "plan op %u: projected shortcut input exceeds buffer %d", i, o.res_buf);
                } else if (o.reserved != 0) {
                    // strided residual (round 6): res_buf is a LARGER map [h2, w2, cout] read at every stride-th pixel; reserved = stride << 12 | h2 << 14 | w2 << 23
                    const int k2 = o.reserved & 0xFFF, st2 = (o.reserved >> 12) & 3, h2 = (o.reserved >> 14) & 0x1FF, w2 = (o.reserved >> 23) & 0x1FF;
                    HSEFR_REQUIRE(k2 == 0 && o.kh == 1 && o.kw == 1 && o.stride == 1 && o.pad_t == 0 && o.pad_l == 0 && o.res_buf >= 0 && st2 >= 1 &&
                                      o.oh == o.h && o.ow == o.w && o.oh * o.ow > 1 && o.ow > 1 && (o.oh - 1) * st2 < h2 && (o.ow - 1) * st2 < w2,
                                  HSEFR_ERR_INVALID, "plan op %u: bad strided residual (stride %d, %dx%d)", i, st2, h2, w2);
                    HSEFR_REQUIRE((uint64_t)h2 * w2 * co * 2 <= bufs[o.res_buf].elems_per_image * bufs[o.res_buf].elem_bytes, HSEFR_ERR_INVALID,
                                  "plan op %u: strided residual exceeds buffer %d", i, o.res_buf);
                }
                break;
            case HSEFR_OP_STEM7X7_POOL_BF16:
                HSEFR_REQUIRE(o.act == HSEFR_ACT_RELU && (o.reserved & ~0x11) == 0 && o.cin == 3 && o.cout == 64, HSEFR_ERR_INVALID,
                              "plan op %u: fused stem + pool needs ReLU, 3 -> 64 channels and pool pads in {0, 1}", i);
                [[fallthrough]];
            case HSEFR_OP_STEM7X7_BF16:
                if (!need(o.w_off, 64 * 256 * 2, "kernel") || !need(o.scale_off, co * 4, "scale") || !need(o.shift_off, co * 4, "shift"))
                    return HSEFR_ERR_INVALID;
                break;
            case HSEFR_OP_PWCONV_PS_GAP:
                HSEFR_REQUIRE(pwconv_ps_gap_supported(0, o.cin, o.cout, o.h * o.w) && o.oh == 1 && o.ow == 1, HSEFR_ERR_UNSUPPORTED,
                              "plan op %u: fused pointwise + global pool on a %dx%d map not covered (33 .. 288 pixels)", i, o.h, o.w);
                if (!need(o.w_off, ci * co * 4, "split rows") || !need(o.scale_off, co * 4, "descale") || !need(o.shift_off, co * 4, "shift"))
                    return HSEFR_ERR_INVALID;
                HSEFR_REQUIRE(o.reserved > 0 && o.reserved <= 24, HSEFR_ERR_INVALID, "plan op %u: a_log2 out of range (reserved %d)", i, o.reserved);
                break;
            case HSEFR_OP_PWCONV_PS_DW:
                HSEFR_REQUIRE(pwconv_ps_dw_supported(0, o.cin, o.cout, o.w, o.h * o.w, o.stride) && o.oh * o.stride == o.h && o.ow * o.stride == o.w &&
                                  (o.stride == 1 ? (o.pad_t == 1 && o.pad_l == 1) : (o.pad_t == 0 && o.pad_l == 0)),
                              HSEFR_ERR_UNSUPPORTED, "plan op %u: fused pointwise + depthwise (stride %d) on a %dx%d map not covered", i, o.stride, o.h, o.w);
                if (!need(o.w_off, ci * co * 4, "split rows") || !need(o.scale_off, co * 4, "descale") || !need(o.shift_off, co * 4, "shift") ||
                    !need(o.w2_off, 11 * co * 4, "depthwise constants"))
                    return HSEFR_ERR_INVALID;
                HSEFR_REQUIRE((o.reserved & 255) > 0 && (o.reserved & 255) <= 24 && (o.reserved >> 8) >= 1 && (o.reserved >> 8) <= 12,
                              HSEFR_ERR_INVALID, "plan op %u: a_log2 / out_log2 out of range (reserved %d)", i, o.reserved);
                break;
            case HSEFR_OP_PWCONV_PS:
                HSEFR_REQUIRE(pwconv_ps_supported(0, o.cin, o.cout), HSEFR_ERR_UNSUPPORTED,
                              "plan op %u: pre-split pointwise cin=%d cout=%d not covered", i, o.cin, o.cout);
                [[fallthrough]];
            case HSEFR_OP_PWCONV_F16S:
                if (!need(o.w_off, ci * co * 4, "split rows") || !need(o.scale_off, co * 4, "descale") || !need(o.shift_off, co * 4, "shift"))
                    return HSEFR_ERR_INVALID;
                HSEFR_REQUIRE(o.w_off != HSEFR_NO_OFFSET && o.scale_off != HSEFR_NO_OFFSET && o.shift_off != HSEFR_NO_OFFSET &&
                                  o.reserved > 0 && o.reserved <= 24,
                              HSEFR_ERR_INVALID, "plan op %u: split-f16 pointwise needs split rows, descale, shift and a_log2 in (0, 24]", i);
                break;
            case HSEFR_OP_STEM3_F16S:
                HSEFR_REQUIRE(stem3_fused_supported(o.cin, 32, o.cout, o.stride, 1, 2, o.kh, o.kw & 15) && (o.reserved & 255) > 0 &&
                                  (o.reserved & 255) <= 12 && ((o.reserved >> 8) & 255) >= 64 - 8 && ((o.reserved >> 8) & 255) <= 64 + 14 &&
                                  o.w_off != HSEFR_NO_OFFSET && o.w2_off != HSEFR_NO_OFFSET && o.shift2_off != HSEFR_NO_OFFSET &&
                                  o.w_off + 7264 * 4 <= h.blob_bytes && o.w2_off + 64 * 128 <= h.blob_bytes &&
                                  o.shift2_off + 128 * 4 <= h.blob_bytes && o.in_buf == HSEFR_BUF_INPUT,
                              HSEFR_ERR_UNSUPPORTED, "plan op %u: bounded fused stem cin=%d cout=%d stride=%d not covered", i, o.cin, o.cout, o.stride);
                break;
            case HSEFR_OP_STEM2_F16S:
                HSEFR_REQUIRE(stem2_fused_supported(o.cin, 32, o.cout, o.stride, 1, 2, o.kh, o.kw & 15) && o.reserved > 0 && o.reserved <= 12 &&
                                  o.w_off != HSEFR_NO_OFFSET && o.w2_off != HSEFR_NO_OFFSET && o.shift2_off != HSEFR_NO_OFFSET &&
                                  o.w_off + 1952 * 4 <= h.blob_bytes && o.in_buf == HSEFR_BUF_INPUT,
                              HSEFR_ERR_UNSUPPORTED, "plan op %u: fused stem+dw2 cin=%d cout=%d stride=%d not covered", i, o.cin, o.cout, o.stride);
                break;
            case HSEFR_OP_STEM_F16S:
#ifndef HSEFR_DEV
                set_error("plan op %u: HSEFR_OP_STEM_F16S (round 1's fused stem) runs on development builds of the library only", i);
                return HSEFR_ERR_UNSUPPORTED;
#else
                HSEFR_REQUIRE(stem_fused_supported(o.cin, 32, o.cout, o.stride, 1, o.kh, o.kw) && o.reserved > 0 && o.reserved <= 12 &&
                                  o.w_off != HSEFR_NO_OFFSET && o.w2_off != HSEFR_NO_OFFSET && o.shift2_off != HSEFR_NO_OFFSET &&
                                  o.w_off + 1248 * 4 <= h.blob_bytes && o.in_buf == HSEFR_BUF_INPUT,
                              HSEFR_ERR_UNSUPPORTED, "plan op %u: fused stem cin=%d cout=%d stride=%d not covered", i, o.cin, o.cout, o.stride);
                break;
#endif
            case HSEFR_OP_DWPW_F16S:
                if (!need(o.w_off, 9 * ci * 4, "depthwise kernel") || !need(o.scale_off, ci * 4, "scale") || !need(o.shift_off, ci * 4, "shift") ||
                    !need(o.w2_off, ci * co * 4, "split rows") || !need(o.shift2_off, 2 * co * 4, "descale | shift"))
                    return HSEFR_ERR_INVALID;
                HSEFR_REQUIRE(dwpw_f16s_supported(o.cin, o.cout, o.stride) && o.reserved > 0 && o.reserved <= 12 &&
                                  o.w_off != HSEFR_NO_OFFSET && o.scale_off != HSEFR_NO_OFFSET && o.shift_off != HSEFR_NO_OFFSET &&
                                  o.w2_off != HSEFR_NO_OFFSET && o.shift2_off != HSEFR_NO_OFFSET,
                              HSEFR_ERR_UNSUPPORTED, "plan op %u: split-f16 fused block cin=%d cout=%d stride=%d a_log2=%d not covered",
                              i, o.cin, o.cout, o.stride, o.reserved);
                break;
            case HSEFR_OP_DWPW_F32:
                if (!need(o.w_off, 9 * ci * 4, "depthwise kernel") || !need(o.scale_off, ci * 4, "scale") || !need(o.shift_off, ci * 4, "shift") ||
                    !need(o.w2_off, ci * co * 4, "pointwise kernel") || !need(o.shift2_off, co * 4, "pointwise shift"))
                    return HSEFR_ERR_INVALID;
                HSEFR_REQUIRE(dwpw_fused_supported(o.cin, o.cout, o.stride, HSEFR_ACT_RELU6, (int)o.act), HSEFR_ERR_UNSUPPORTED,
                              "plan op %u: fused depthwise-pointwise block cin=%d cout=%d not covered", i, o.cin, o.cout);
                break;
            default:
                set_error("plan op %u: unknown kind %u", i, o.kind);
                return HSEFR_ERR_UNSUPPORTED;
        }
    }
    // launch-level fusions (hsefr_op_flags): the pattern behind a flagged op must be exactly the one its fused launch computes
    for (uint32_t i = 0; i < h.n_ops; ++i) {
        const hsefr_plan_op& o = ops[i];
        HSEFR_REQUIRE((o.flags & ~(HSEFR_OPF_PAIR_NEXT | HSEFR_OPF_HEADS | HSEFR_OPF_OUT_SUB2)) == 0, HSEFR_ERR_INVALID, "plan op %u: unknown flags 0x%x", i, o.flags);
        HSEFR_REQUIRE(!(o.flags & HSEFR_OPF_OUT_SUB2) || (o.flags & HSEFR_OPF_PAIR_NEXT), HSEFR_ERR_INVALID, "plan op %u: OUT_SUB2 without PAIR_NEXT", i);
        if (o.flags & HSEFR_OPF_PAIR_NEXT) {
            HSEFR_REQUIRE(i + 1 < h.n_ops, HSEFR_ERR_INVALID, "plan op %u: PAIR_NEXT on the last op", i);
            const hsefr_plan_op& b = ops[i + 1];
            const bool proj = o.w2_off != HSEFR_NO_OFFSET;
            const int c2 = proj ? (o.reserved & 0xFFF) : 0, st2 = (o.reserved >> 12) & 3, h2 = (o.reserved >> 14) & 0x1FF, w2 = (o.reserved >> 23) & 0x1FF;
            const bool sub2 = (o.flags & HSEFR_OPF_OUT_SUB2) != 0;      // the first output is stored at even rows / columns only; the second op reads all of it
            HSEFR_REQUIRE(o.kind == HSEFR_OP_CONV_BF16 && b.kind == HSEFR_OP_CONV_BF16 && o.kh == 1 && o.kw == 1 && o.stride == 1 && o.pad_t == 0 &&
                              o.pad_l == 0 && (sub2 ? (o.oh == (o.h + 1) / 2 && o.ow == (o.w + 1) / 2 && o.h > 1 && o.w > 1) : (o.oh == o.h && o.ow == o.w)) &&
                              b.kh == 1 && b.kw == 1 && b.stride == 1 && b.pad_t == 0 && b.pad_l == 0 &&
                              b.h == o.h && b.w == o.w && b.oh == b.h && b.ow == b.w && b.cin == o.cout && b.in_buf == o.out_buf &&
                              b.res_buf == HSEFR_BUF_NONE && b.w2_off == HSEFR_NO_OFFSET && b.flags == 0 && o.res_buf >= 0 &&
                              (!proj || (st2 == 1 && h2 == o.h && w2 == o.w)),
                          HSEFR_ERR_INVALID, "plan op %u: PAIR_NEXT needs two 1x1 stride-1 bf16 convolutions at the same pixels, the first with a residual "
                          "or a same-pixel projected shortcut, the second reading the first", i);
            HSEFR_REQUIRE(b.out_buf != o.in_buf && b.out_buf != o.res_buf && b.out_buf != o.out_buf, HSEFR_ERR_INVALID,
                          "plan op %u: PAIR_NEXT: the second output (buffer %d) aliases an operand of the first", i, b.out_buf);
            HSEFR_REQUIRE(conv1x1_pair_bf16_shape_supported(o.cin, o.cout, b.cout, c2), HSEFR_ERR_UNSUPPORTED,
                          "plan op %u: PAIR_NEXT %d -> %d -> %d (projection from %d) not covered", i, o.cin, o.cout, b.cout, c2);
        }
        if (o.flags & HSEFR_OPF_HEADS) {
            HSEFR_REQUIRE(i + 3 < h.n_ops, HSEFR_ERR_INVALID, "plan op %u: HEADS needs three ops behind it", i);
            const hsefr_plan_op &a = ops[i + 1], &sm = ops[i + 2], &g = ops[i + 3];
            HSEFR_REQUIRE(o.kind == HSEFR_OP_DENSE && o.act == HSEFR_ACT_RELU && o.cout == 256 && o.cin > 0 && o.cin % 256 == 0 && o.cin <= 2048 &&
                              o.shift_off != HSEFR_NO_OFFSET && a.kind == HSEFR_OP_DENSE && a.act == HSEFR_ACT_NONE && a.cin == 256 && a.cout >= 1 &&
                              a.cout <= 128 && a.in_buf == o.out_buf && a.shift_off != HSEFR_NO_OFFSET && sm.kind == HSEFR_OP_SOFTMAX &&
                              sm.in_buf == a.out_buf && sm.cout == a.cout && g.kind == HSEFR_OP_DENSE && g.act == HSEFR_ACT_SIGMOID && g.cin == 256 &&
                              g.cout == 1 && g.in_buf == o.out_buf && g.shift_off != HSEFR_NO_OFFSET && a.flags == 0 && sm.flags == 0 && g.flags == 0,
                          HSEFR_ERR_INVALID, "plan op %u: HEADS needs DENSE k -> 256 + ReLU, DENSE 256 -> a (<= 128) + bias, SOFTMAX, DENSE 256 -> 1 + sigmoid", i);
            const int outs[4] = {o.out_buf, a.out_buf, sm.out_buf, g.out_buf};
            for (int x = 0; x < 4; ++x) {
                HSEFR_REQUIRE(outs[x] != o.in_buf, HSEFR_ERR_INVALID, "plan op %u: HEADS: an output aliases the pooled features", i);
                for (int y = x + 1; y < 4; ++y)
                    HSEFR_REQUIRE(outs[x] != outs[y], HSEFR_ERR_INVALID, "plan op %u: HEADS: two of the four tensors share buffer %d", i, outs[x]);
            }
        }
    }
    for (int s = 0; s < HSEFR_N_OUTPUT_SLOTS; ++s) {
        HSEFR_REQUIRE(h.out_buffer[s] == HSEFR_BUF_NONE || (h.out_buffer[s] >= 0 && h.out_buffer[s] < (int)h.n_buffers),
                      HSEFR_ERR_INVALID, "plan: output slot %d names buffer %d", s, h.out_buffer[s]);
        if (h.out_buffer[s] == HSEFR_BUF_NONE) continue;
        // hsefr_engine_forward lets the producing kernel write the CALLER's [n, out_elems] tensor in place of this buffer:
        // that is only sound if the buffer has exactly one producer (never recycled for another tensor) and holds exactly
        // out_elems fp32 values per image -- a hand-made plan that breaks either would overrun the caller's memory.
        const int b = h.out_buffer[s];
        uint32_t producers = 0;
        for (uint32_t i = 0; i < h.n_ops; ++i) producers += (ops[i].out_buf == b);
        HSEFR_REQUIRE(producers == 1, HSEFR_ERR_INVALID, "plan: output slot %d: buffer %d has %u producing ops (must be exactly 1)", s, b,
                      producers);
        HSEFR_REQUIRE((uint64_t)bufs[b].elems_per_image * bufs[b].elem_bytes == (uint64_t)h.out_elems[s] * 4, HSEFR_ERR_INVALID,
                      "plan: output slot %d: buffer %d holds %llu x %u bytes per image, the slot declares %u fp32 elements", s, b,
                      (unsigned long long)bufs[b].elems_per_image, (unsigned)bufs[b].elem_bytes, h.out_elems[s]);
    }
    return HSEFR_OK;
}

namespace hsefr {
#ifdef HSEFR_STEM_STAMPS
static unsigned long long* g_stamp_buf = nullptr;
unsigned long long* stamp_buffer(hipStream_t s) {
    if (!g_stamp_buf && hipMalloc((void**)&g_stamp_buf, 512 * 4 * 10 * 8) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(g_stamp_buf, 0, 512 * 4 * 10 * 8, s);
    return g_stamp_buf;
}
int read_stem_stamps(void* host_out, size_t bytes) {
    HSEFR_REQUIRE(g_stamp_buf && bytes <= 512 * 4 * 10 * 8, HSEFR_ERR_INVALID, "read_stem_stamps: nothing recorded / too many bytes");
    HSEFR_HIP_CHECK(hipMemcpy(host_out, g_stamp_buf, bytes, hipMemcpyDeviceToHost));
    return HSEFR_OK;
}
#else
int read_stem_stamps(void* host_out, size_t bytes) {
    (void)host_out; (void)bytes;
    set_error("read_stem_stamps: library built without -DHSEFR_STEM_STAMPS");
    return HSEFR_ERR_UNSUPPORTED;
}
#endif
// ---- route probe (common.h HSEFR_LAUNCH) ----
static thread_local std::string* g_route_sink = nullptr;
bool route_probe() { return g_route_sink != nullptr; }
// kernel handle address -> demangled kernel name with its template arguments, from the library's OWN symbol table (the handles are local
// symbols: hidden visibility, anonymous namespaces -- dladdr does not see them, .symtab does); built once, on the first describe
static const std::map<uintptr_t, std::string>& stub_names() {
    static const std::map<uintptr_t, std::string> table = [] {
        std::map<uintptr_t, std::string> m;
        Dl_info info;
        if (!dladdr(reinterpret_cast<const void*>(&stub_names), &info) || !info.dli_fname) return m;
        FILE* f = fopen(info.dli_fname, "rb");
        if (!f) return m;
        std::vector<unsigned char> img;
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz > 0) { img.resize((size_t)sz); if (fread(img.data(), 1, img.size(), f) != img.size()) img.clear(); }
        fclose(f);
        if (img.size() < sizeof(Elf64_Ehdr)) return m;
        const Elf64_Ehdr* eh = reinterpret_cast<const Elf64_Ehdr*>(img.data());
        if (memcmp(eh->e_ident, ELFMAG, SELFMAG) != 0 || eh->e_shentsize != sizeof(Elf64_Shdr) ||
            eh->e_shoff + (uint64_t)eh->e_shnum * sizeof(Elf64_Shdr) > img.size()) return m;
        const Elf64_Shdr* sh = reinterpret_cast<const Elf64_Shdr*>(img.data() + eh->e_shoff);
        for (int i = 0; i < eh->e_shnum; ++i) {
            if (sh[i].sh_type != SHT_SYMTAB || sh[i].sh_link >= eh->e_shnum) continue;
            const Elf64_Shdr& st = sh[sh[i].sh_link];
            if (sh[i].sh_offset + sh[i].sh_size > img.size() || st.sh_offset + st.sh_size > img.size()) continue;
            const Elf64_Sym* sym = reinterpret_cast<const Elf64_Sym*>(img.data() + sh[i].sh_offset);
            const char* str = reinterpret_cast<const char*>(img.data() + st.sh_offset);
            for (size_t k = 0; k < sh[i].sh_size / sizeof(Elf64_Sym); ++k) {
                // (a kernel's host-side HANDLE -- what `&kernel` evaluates to in host code -- is a data symbol with the kernel's own mangled name)
                if (ELF64_ST_TYPE(sym[k].st_info) != STT_OBJECT || sym[k].st_name >= st.sh_size) continue;
                const char* nm = str + sym[k].st_name;
                if (strncmp(nm, "_ZN5hsefr", 9) != 0 || !strstr(nm, "_kernel")) continue;
                int status = 0;
                char* dm = abi::__cxa_demangle(nm, nullptr, nullptr, &status);
                std::string t = (status == 0 && dm) ? dm : nm;
                free(dm);
                size_t cut = std::string::npos, depth = 0;
                for (size_t q = 0; q < t.size(); ++q) {       // the '(' that opens the parameter list: the first one outside <...> after the name
                    if (t[q] == '<') ++depth;
                    else if (t[q] == '>') --depth;
                    else if (t[q] == '(' && depth == 0 && q > 0 && t.compare(q, 21, "(anonymous namespace)") != 0) { cut = q; break; }
                }
                if (cut != std::string::npos) t.erase(cut);
                for (const char* ns : {"void ", "hsefr::", "(anonymous namespace)::", "__device_stub__"})
                    for (size_t q; (q = t.find(ns)) != std::string::npos;) t.erase(q, strlen(ns));
                m[(uintptr_t)info.dli_fbase + sym[k].st_value] = t;
            }
        }
        return m;
    }();
    return table;
}
void route_record(const void* host_stub, const char* expr) {
    if (!g_route_sink) return;
    const auto& tab = stub_names();
    const auto it = tab.find((uintptr_t)host_stub);
    std::string t = it != tab.end() ? it->second : std::string(expr ? expr : "?");
    if (!g_route_sink->empty() && g_route_sink->back() != '\t') *g_route_sink += " + ";
    *g_route_sink += t;
}
static thread_local int g_sweep_reverse = 0;    // set per op by the forward running on THIS host thread, read by its launchers
int sweep_reverse() { return g_sweep_reverse; }
void set_sweep_reverse(int v) { g_sweep_reverse = v; }
}  // namespace hsefr
HSEFR_KNOB(g_sweep_alternate, 1);   // dev builds: 0 turns the alternation off (A/B timing)
HSEFR_KNOB(g_stem5, 1);             // dev builds: 0 = stem4_fused.hip (round 3's patch kernel) where stem5_stream.hip covers the shape (A/B timing)
HSEFR_KNOB(g_heads_off, 0);         // dev builds: 1 = the four head launches also where a DENSE op carries HSEFR_OPF_HEADS (A/B timing)
HSEFR_KNOB(g_stem4, 1);             // dev builds: 0 = stem3_fused.hip also where stem4_fused.hip covers the shape (A/B timing)

#pragma GCC visibility push(default)   // the library is built with -fvisibility=hidden: the C ABI below is ALL it exports
extern "C" {

int hsefr_version(void) { return HSEFR_VERSION; }

const char* hsefr_last_error_string(void) { return g_err; }

#ifdef HSEFR_DEV
int hsefr_debug_set(const char* key, int value) {
    HSEFR_REQUIRE(key, HSEFR_ERR_INVALID, "debug_set: null key");
    if (!strcmp(key, "pw_tile")) { set_pw_tile(value); return HSEFR_OK; }
    if (!strcmp(key, "pws_tile")) { set_pws_tile(value); return HSEFR_OK; }
    if (!strcmp(key, "ps_mb")) { set_ps_mb(value); return HSEFR_OK; }
    if (!strcmp(key, "ps_grid")) { set_ps_grid(value); return HSEFR_OK; }
    if (!strcmp(key, "psdw_mode")) { set_psdw_mode(value); return HSEFR_OK; }
    if (!strcmp(key, "cd_rb")) { set_cd_rb(value); return HSEFR_OK; }
    if (!strcmp(key, "cd_off")) { set_cd_off(value); return HSEFR_OK; }
    if (!strcmp(key, "w3_off")) { set_w3_off(value); return HSEFR_OK; }
    if (!strcmp(key, "w2_off")) { set_w2_off(value); return HSEFR_OK; }
    if (!strcmp(key, "w4_off")) { set_w4_off(value); return HSEFR_OK; }
    if (!strcmp(key, "w4_bres")) { set_w4_bres(value); return HSEFR_OK; }
    if (!strcmp(key, "nn1_y_mb")) { set_nn1_y_mb(value); return HSEFR_OK; }
    if (!strcmp(key, "c11")) { set_c11(value); return HSEFR_OK; }
    if (!strcmp(key, "c11_tile")) { set_c11_tile(value); return HSEFR_OK; }
    if (!strcmp(key, "c11_bres")) { set_c11_bres(value); return HSEFR_OK; }
    if (!strcmp(key, "c11_adv")) { set_c11_adv(value); return HSEFR_OK; }
    if (!strcmp(key, "stem4_grid")) { set_stem4_grid(value); return HSEFR_OK; }
    if (!strcmp(key, "stem4")) { g_stem4 = value; return HSEFR_OK; }
    if (!strcmp(key, "stem5")) { g_stem5 = value; return HSEFR_OK; }
    if (!strcmp(key, "stem5_grid")) { set_stem5_grid(value); return HSEFR_OK; }
    if (!strcmp(key, "stem5_segs")) { set_stem5_segs(value); return HSEFR_OK; }
    if (!strcmp(key, "pair_off")) { set_pair_off(value); return HSEFR_OK; }
    if (!strcmp(key, "stem7s")) { set_stem7s(value); return HSEFR_OK; }
    if (!strcmp(key, "pair_ablate")) { set_pair_ablate(value); return HSEFR_OK; }
    if (!strcmp(key, "pair_nt")) { set_pair_nt(value); return HSEFR_OK; }
    if (!strcmp(key, "heads_off")) { g_heads_off = value; return HSEFR_OK; }
    if (!strcmp(key, "dw_look")) { set_dw_look(value); return HSEFR_OK; }
    if (!strcmp(key, "dw_look2")) { set_dw_look2(value); return HSEFR_OK; }
    if (!strcmp(key, "sweep_alternate")) { g_sweep_alternate = value; return HSEFR_OK; }
    if (!strcmp(key, "clock_mode")) { set_clock_mode(value); return HSEFR_OK; }
    if (!strcmp(key, "dwpws_tw")) { set_dwpws_tw(value); return HSEFR_OK; }
    if (!strcmp(key, "dwpws_bn")) { set_dwpws_bn(value); return HSEFR_OK; }
    if (!strcmp(key, "pw_ablate")) { set_pw_ablate(value); return HSEFR_OK; }
    if (!strcmp(key, "pw_dma")) { set_pw_dma(value); return HSEFR_OK; }
    if (!strcmp(key, "dw_th")) { set_dw_th(value); return HSEFR_OK; }
    if (!strcmp(key, "dw_variant")) { set_dw_variant(value); return HSEFR_OK; }
    if (!strcmp(key, "copy_variant")) { set_copy_variant(value); return HSEFR_OK; }
    if (!strcmp(key, "c3_impl")) { set_c3_impl(value); return HSEFR_OK; }
    set_error("debug_set: unknown key %s", key);
    return HSEFR_ERR_INVALID;
}

int hsefr_debug_copy(const void* d_src, void* d_dst, size_t bytes, hsefr_stream_t stream) {
    HSEFR_REQUIRE(bytes == 0 || (d_src && d_dst), HSEFR_ERR_INVALID, "debug_copy: null pointer");
    return launch_copy(d_src, d_dst, bytes, (hipStream_t)stream);
}

int hsefr_debug_read_stamps(int kernel, void* host_out, size_t bytes) {
    switch (kernel) {
        case HSEFR_STAMPS_PWS: return read_pws_stamps(host_out, bytes);
        case HSEFR_STAMPS_STEM: return read_stem_stamps(host_out, bytes);
        case HSEFR_STAMPS_PS: return read_ps_stamps(host_out, bytes);
        case HSEFR_STAMPS_CD: return read_cd_stamps(host_out, bytes);
        case HSEFR_STAMPS_C11: return read_c11_stamps(host_out, bytes);
        case HSEFR_STAMPS_W4: return read_w4_stamps(host_out, bytes);
        case HSEFR_STAMPS_W2: return read_w2_stamps(host_out, bytes);
        case HSEFR_STAMPS_W3: return read_w3_stamps(host_out, bytes);
        case HSEFR_STAMPS_S7: return read_s7_stamps(host_out, bytes);
        default: set_error("debug_read_stamps: unknown kernel id %d", kernel); return HSEFR_ERR_INVALID;
    }
}

int hsefr_debug_clock_probe(unsigned long long* d_out, int blocks, int iters, hsefr_stream_t stream) {
    return launch_clock_probe(d_out, blocks, iters, (hipStream_t)stream);
}
#endif  // HSEFR_DEV

// size and structure of a plan blob, then validate_plan: everything hsefr_engine_create checks before it touches the device
static int check_plan_blob(const void* plan, size_t plan_bytes, hsefr_plan_header& h, std::vector<hsefr_plan_buffer>& bufs,
                           std::vector<hsefr_plan_op>& ops, const char*& blob) {
    HSEFR_REQUIRE(plan, HSEFR_ERR_INVALID, "plan: null pointer");
    HSEFR_REQUIRE(plan_bytes >= sizeof(hsefr_plan_header), HSEFR_ERR_INVALID, "engine_create: plan too short");
    memcpy(&h, plan, sizeof(h));
    HSEFR_REQUIRE(h.magic == HSEFR_PLAN_MAGIC && h.version == 2, HSEFR_ERR_INVALID, "engine_create: bad plan magic/version");
    // (64-bit sums of 32-bit counts: no overflow; a blob_bytes field near 2^64 cannot equal plan_bytes minus the tables)
    const unsigned long long tables = sizeof(h) + (unsigned long long)h.n_buffers * sizeof(hsefr_plan_buffer) +
                                      (unsigned long long)h.n_ops * sizeof(hsefr_plan_op);
    HSEFR_REQUIRE(tables <= plan_bytes && h.blob_bytes == plan_bytes - tables, HSEFR_ERR_INVALID,
                  "engine_create: plan is %zu bytes, header implies %llu + %llu", plan_bytes, tables, (unsigned long long)h.blob_bytes);
    HSEFR_REQUIRE(h.n_buffers > 0 && h.n_ops > 0, HSEFR_ERR_INVALID, "engine_create: plan without %s", h.n_ops ? "buffers" : "ops");
    const char* p = (const char*)plan + sizeof(h);
    try {
        bufs.resize(h.n_buffers);
        ops.resize(h.n_ops);
    } catch (const std::bad_alloc&) {
        set_error("engine_create: out of host memory for %u buffers / %u ops", h.n_buffers, h.n_ops);
        return HSEFR_ERR_NOMEM;
    }
    memcpy(bufs.data(), p, h.n_buffers * sizeof(hsefr_plan_buffer));
    p += h.n_buffers * sizeof(hsefr_plan_buffer);
    memcpy(ops.data(), p, h.n_ops * sizeof(hsefr_plan_op));
    blob = p + h.n_ops * sizeof(hsefr_plan_op);
    return validate_plan(h, bufs.data(), ops.data());
}

int hsefr_plan_validate(const void* plan, size_t plan_bytes) {
    hsefr_plan_header h;
    std::vector<hsefr_plan_buffer> bufs;
    std::vector<hsefr_plan_op> ops;
    const char* blob = nullptr;
    return check_plan_blob(plan, plan_bytes, h, bufs, ops, blob);
}

int hsefr_engine_create(const void* plan, size_t plan_bytes, int max_batch, hsefr_engine** out) {
    HSEFR_REQUIRE(plan && out, HSEFR_ERR_INVALID, "engine_create: null argument");
    *out = nullptr;
    HSEFR_REQUIRE(max_batch > 0, HSEFR_ERR_INVALID, "engine_create: max_batch=%d", max_batch);
    hsefr_engine* e = new (std::nothrow) hsefr_engine();
    HSEFR_REQUIRE(e, HSEFR_ERR_NOMEM, "engine_create: out of host memory");
    const char* p = nullptr;
    int rc = check_plan_blob(plan, plan_bytes, e->hdr, e->bufs, e->ops, p);
    if (rc != HSEFR_OK) { delete e; return rc; }
    const hsefr_plan_header& h = e->hdr;
    e->max_batch = max_batch;

    auto fail = [&](int code) { hsefr_engine_destroy(e); return code; };
    if (hipGetDevice(&e->device) != hipSuccess) { set_error("engine_create: hipGetDevice failed"); return fail(HSEFR_ERR_HIP); }
    if (h.blob_bytes) {
        if (hipMalloc((void**)&e->d_blob, h.blob_bytes) != hipSuccess) {
            set_error("engine_create: hipMalloc(%llu) for weights failed", (unsigned long long)h.blob_bytes);
            return fail(HSEFR_ERR_NOMEM);
        }
        if (hipMemcpy(e->d_blob, p, h.blob_bytes, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("engine_create: weight upload failed");
            return fail(HSEFR_ERR_HIP);
        }
        e->device_bytes += h.blob_bytes;
    }
    if (hipMalloc((void**)&e->d_overflow, 16) != hipSuccess || hipMemset(e->d_overflow, 0, 16) != hipSuccess) {
        set_error("engine_create: hipMalloc for the overflow flag failed");
        return fail(HSEFR_ERR_NOMEM);
    }
    e->d_derived.assign(e->ops.size(), nullptr);
    for (size_t i = 0; i < e->ops.size(); ++i) {
        const hsefr_plan_op& o = e->ops[i];
        if (o.kind != HSEFR_OP_STEM7X7_POOL_BF16) continue;
        if (hipMalloc(&e->d_derived[i], STEM7S_WFRAG_BYTES) != hipSuccess) {
            set_error("engine_create: hipMalloc for the stem's fragment-ordered weights failed");
            return fail(HSEFR_ERR_NOMEM);
        }
        e->device_bytes += STEM7S_WFRAG_BYTES;
        if (int rc = stem7s_reorder_weights(blob_ptr(e, o.w_off), e->d_derived[i], nullptr)) return fail(rc);
        if (hipStreamSynchronize(nullptr) != hipSuccess) { set_error("engine_create: re-ordering the stem's weights failed"); return fail(HSEFR_ERR_HIP); }
    }
    e->d_bufs.assign(h.n_buffers, nullptr);
    for (uint32_t i = 0; i < h.n_buffers; ++i) {
        const size_t bytes = (size_t)e->bufs[i].elems_per_image * e->bufs[i].elem_bytes * max_batch;
        if (hipMalloc(&e->d_bufs[i], bytes ? bytes : 16) != hipSuccess) {
            set_error("engine_create: hipMalloc(%zu) for activation buffer %u failed", bytes, i);
            return fail(HSEFR_ERR_NOMEM);
        }
        e->device_bytes += bytes;
    }
    *out = e;
    return HSEFR_OK;
}

size_t hsefr_engine_workspace_bytes(const hsefr_engine* e) { return e ? e->device_bytes : 0; }
int hsefr_engine_max_batch(const hsefr_engine* e) { return e ? e->max_batch : 0; }

void* hsefr_engine_buffer(hsefr_engine* e, int buffer) {
    if (!e || buffer < 0 || buffer >= (int)e->d_bufs.size()) return nullptr;
    return e->d_bufs[buffer];
}

int hsefr_engine_copy_buffer(hsefr_engine* e, int buffer, void* d_dst, size_t bytes, hsefr_stream_t stream) {
    HSEFR_REQUIRE(e && d_dst, HSEFR_ERR_INVALID, "copy_buffer: null argument");
    HSEFR_REQUIRE(buffer >= 0 && buffer < (int)e->d_bufs.size(), HSEFR_ERR_INVALID, "copy_buffer: buffer %d", buffer);
    const size_t cap = (size_t)e->bufs[buffer].elems_per_image * e->bufs[buffer].elem_bytes * e->max_batch;
    HSEFR_REQUIRE(bytes <= cap, HSEFR_ERR_SHAPE, "copy_buffer: %zu bytes exceed buffer %d (%zu)", bytes, buffer, cap);
    HSEFR_HIP_CHECK(hipMemcpyAsync(d_dst, e->d_bufs[buffer], bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return HSEFR_OK;
}

int hsefr_engine_set_profiling(hsefr_engine* e, int depth) {
    HSEFR_REQUIRE(e && depth >= 0, HSEFR_ERR_INVALID, "set_profiling: bad argument");
    for (auto ev : e->events) (void)hipEventDestroy(ev);
    e->events.clear();
    e->prof_depth = 0;
    e->prof_calls = 0;
    if (depth > 0) {
        e->events.resize((size_t)depth * (e->ops.size() + 1));
        for (auto& ev : e->events) HSEFR_HIP_CHECK(hipEventCreate(&ev));
        e->prof_depth = depth;
    }
    return HSEFR_OK;
}

long long hsefr_engine_profiled_calls(const hsefr_engine* e) { return e ? e->prof_calls : 0; }

int hsefr_engine_op_times_ms(hsefr_engine* e, int slot, float* ms, int n_ops) {
    HSEFR_REQUIRE(e && ms, HSEFR_ERR_INVALID, "op_times: null argument");
    HSEFR_REQUIRE(e->prof_depth > 0 && slot >= 0 && slot < e->prof_depth && slot < e->prof_calls, HSEFR_ERR_INVALID,
                  "op_times: slot %d has no profiled forward (depth %d, calls %lld)", slot, e->prof_depth, e->prof_calls);
    HSEFR_REQUIRE(n_ops == (int)e->ops.size(), HSEFR_ERR_INVALID, "op_times: expected %zu ops", e->ops.size());
    hipEvent_t* ev = e->events.data() + (size_t)slot * (e->ops.size() + 1);
    HSEFR_HIP_CHECK(hipEventSynchronize(ev[n_ops]));
    for (int i = 0; i < n_ops; ++i) HSEFR_HIP_CHECK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
    return HSEFR_OK;
}

// Launch the needed ops of the plan for a batch of n on stream s (plain launches: also what a graph capture records).
static int run_ops(hsefr_engine* e, const std::vector<void*>& tab, const void* d_input, int n, const std::vector<char>& needed,
                   hipStream_t s, hipEvent_t* pev, bool input_u8 = false) {
    const bool prof = pev != nullptr;
    unsigned launches = 0;     // (a launch that covers several ops -- hsefr_op_flags -- counts once: the op behind it must still sweep the other way)
    if (prof) HSEFR_HIP_CHECK(hipEventRecord(pev[0], s));
    for (size_t i = 0; i < e->ops.size(); ++i) {
        const hsefr_plan_op& o = e->ops[i];
        if (!needed[i]) {
            if (prof) HSEFR_HIP_CHECK(hipEventRecord(pev[i + 1], s));
            continue;
        }
        const void* in = buf_ptr(tab, o.in_buf, d_input);
        void* out = buf_ptr(tab, o.out_buf, d_input);
        int rc = HSEFR_OK;
        size_t covered = 0;       // ops behind this one that its launch computes as well (hsefr_op_flags)
        set_sweep_reverse(g_sweep_alternate ? (int)(launches++ & 1) : 0);   // consecutive LAUNCHES sweep in opposite directions (common.h)
        const bool sub2 = (o.flags & HSEFR_OPF_OUT_SUB2) != 0;
        const bool pair_ok = (o.flags & HSEFR_OPF_PAIR_NEXT) && needed[i + 1] &&
                             conv1x1_pair_bf16_supported((long long)n * o.h * o.w, o.cin, o.cout, e->ops[i + 1].cout, o.w2_off != HSEFR_NO_OFFSET ? (o.reserved & 0xFFF) : 0);
        HSEFR_REQUIRE(!sub2 || pair_ok, HSEFR_ERR_UNSUPPORTED, "forward: op %zu stores its output at every second pixel (OUT_SUB2) and cannot run without the pair launch", i);
        if (pair_ok) {
            // increase (+ residual | + projected shortcut) -> the next block's reduce in one launch (csrc/conv1x1_pair_bf16.hip; the pattern
            // was checked by validate_plan): both tensors are written, the second product reads the first from registers
            const hsefr_plan_op& b = e->ops[i + 1];
            const bool proj = o.w2_off != HSEFR_NO_OFFSET;
            const float* ssp = (const float*)blob_ptr(e, o.shift2_off);
            rc = launch_conv1x1_pair_bf16(in, blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off), (const float*)blob_ptr(e, o.shift_off),
                                          proj ? nullptr : tab[o.res_buf], proj ? tab[o.res_buf] : nullptr, proj ? blob_ptr(e, o.w2_off) : nullptr,
                                          proj ? ssp : nullptr, proj ? ssp + o.cout : nullptr, out, blob_ptr(e, b.w_off),
                                          (const float*)blob_ptr(e, b.scale_off), (const float*)blob_ptr(e, b.shift_off), tab[b.out_buf],
                                          (long long)n * o.h * o.w, o.cin, o.cout, b.cout, proj ? (o.reserved & 0xFFF) : 0, o.act, b.act, s, sub2 ? 2 : 1, o.h, o.w);
            covered = 1;
        } else if ((o.flags & HSEFR_OPF_HEADS) && !g_heads_off && needed[i + 1] && needed[i + 2] && needed[i + 3]) {
            // the age / gender heads in one launch (csrc/pool_dense.hip): hidden, logits, probabilities and the gender sigmoid are all written
            const hsefr_plan_op &a = e->ops[i + 1], &sm = e->ops[i + 2], &g = e->ops[i + 3];
            rc = launch_heads_fused((const float*)in, (const float*)blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.shift_off),
                                    (const float*)blob_ptr(e, a.w_off), (const float*)blob_ptr(e, a.shift_off), (const float*)blob_ptr(e, g.w_off),
                                    (const float*)blob_ptr(e, g.shift_off), (float*)out, (float*)tab[a.out_buf], (float*)tab[sm.out_buf],
                                    (float*)tab[g.out_buf], n, o.cin, a.cout, s);
            covered = 3;
        } else
        switch (o.kind) {
            case HSEFR_OP_CONV_C3:
                rc = launch_conv_c3((const float*)in, (const float*)blob_ptr(e, o.w_off),
                                    (const float*)blob_ptr(e, o.shift_off), (float*)out, n, o.h, o.w, o.kh, o.kw,
                                    o.stride, o.pad_t, o.pad_l, o.oh, o.ow, o.cout, o.act, s);
                break;
            case HSEFR_OP_DWCONV3X3:
                if (o.reserved > 0)
                    rc = launch_dwconv3x3_split((const float*)in, (const float*)blob_ptr(e, o.w_off),
                                                (const float*)blob_ptr(e, o.scale_off), (const float*)blob_ptr(e, o.shift_off),
                                                out, n, o.h, o.w, o.cin, o.stride, o.pad_t, o.pad_l, o.oh, o.ow, o.act, o.reserved, s);
                else
                    rc = launch_dwconv3x3((const float*)in, (const float*)blob_ptr(e, o.w_off),
                                          (const float*)blob_ptr(e, o.scale_off), (const float*)blob_ptr(e, o.shift_off),
                                          (float*)out, n, o.h, o.w, o.cin, o.stride, o.pad_t, o.pad_l, o.oh, o.ow, o.act, s);
                break;
            case HSEFR_OP_PWCONV_PS_GAP:
                rc = launch_pwconv_ps_gap(in, blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off), (const float*)blob_ptr(e, o.shift_off),
                                          (float*)out, (long long)n * o.h * o.w, o.cin, o.cout, o.act, o.h * o.w, s);
                break;
            case HSEFR_OP_PWCONV_PS_DW:
                rc = launch_pwconv_ps_dw(in, blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off), (const float*)blob_ptr(e, o.shift_off),
                                         (const float*)blob_ptr(e, o.w2_off), out, (long long)n * o.h * o.w, o.cin, o.cout, o.act, o.w,
                                         o.h * o.w, o.stride, o.reserved >> 8, s);
                break;
            case HSEFR_OP_PWCONV_PS:
                rc = launch_pwconv_ps(in, blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off),
                                      (const float*)blob_ptr(e, o.shift_off), (float*)out, (long long)n * o.h * o.w, o.cin, o.cout,
                                      o.act, s);
                break;
            case HSEFR_OP_PWCONV_F32:
                rc = launch_pwconv_f32((const float*)in, (const float*)blob_ptr(e, o.w_off),
                                       (const float*)blob_ptr(e, o.shift_off), (float*)out,
                                       (long long)n * o.h * o.w, o.cin, o.cout, o.act, s);
                break;
            case HSEFR_OP_PWCONV_F16S:
                rc = launch_pwconv_f16s((const float*)in, blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off),
                                        (const float*)blob_ptr(e, o.shift_off), (float*)out, (long long)n * o.h * o.w,
                                        o.cin, o.cout, o.reserved, o.act, s);
                break;
            case HSEFR_OP_GAP:
                rc = launch_gap((const float*)in, (float*)out, n, o.h * o.w, o.cin, s);
                break;
            case HSEFR_OP_DENSE:
                rc = launch_dense((const float*)in, (const float*)blob_ptr(e, o.w_off),
                                  (const float*)blob_ptr(e, o.shift_off), (float*)out, n, o.cin, o.cout, o.act, s);
                break;
            case HSEFR_OP_SOFTMAX:
                rc = launch_softmax((const float*)in, (float*)out, n, o.cout, s);
                break;
            case HSEFR_OP_CONV_BF16:
                if (o.w2_off != HSEFR_NO_OFFSET) {      // increase layer + projected shortcut in one launch (csrc/conv1x1_bf16.hip, PROJ)
                    const float* ss2 = (const float*)blob_ptr(e, o.shift2_off);
                    rc = launch_conv1x1_proj_bf16(in, blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off),
                                                  (const float*)blob_ptr(e, o.shift_off), tab[o.res_buf], blob_ptr(e, o.w2_off), ss2, ss2 + o.cout,
                                                  out, n, o.oh, o.ow, o.cin, o.cout, o.reserved & 0xFFF, (o.reserved >> 12) & 3,
                                                  (o.reserved >> 14) & 0x1FF, (o.reserved >> 23) & 0x1FF, o.act, s);
                    break;
                }
                if (o.res_buf >= 0 && o.reserved != 0) {      // the residual is a stride view of a larger map (csrc/conv1x1_bf16.hip, rs_stride)
                    rc = launch_conv1x1_sres_bf16(in, blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off), (const float*)blob_ptr(e, o.shift_off),
                                                  tab[o.res_buf], out, n, o.oh, o.ow, o.cin, o.cout, (o.reserved >> 12) & 3, (o.reserved >> 14) & 0x1FF,
                                                  (o.reserved >> 23) & 0x1FF, o.act, s);
                    break;
                }
                rc = launch_conv_bf16(in, blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off),
                                      (const float*)blob_ptr(e, o.shift_off),
                                      o.res_buf >= 0 ? tab[o.res_buf] : nullptr, out, n, o.h, o.w, o.cin, o.oh,
                                      o.ow, o.cout, o.kh, o.kw, o.stride, o.pad_t, o.pad_l, o.act, s);
                break;
            case HSEFR_OP_CONV_F32:
                if (conv_f32_mfma_supported(o.cin, o.cout)) {       // exact fp32 on the fp32 matrix pipe (csrc/conv_f32_mfma.hip)
                    rc = launch_conv_f32_mfma((const float*)in, (const float*)blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off),
                                              (const float*)blob_ptr(e, o.shift_off), o.res_buf >= 0 ? (const float*)tab[o.res_buf] : nullptr,
                                              (float*)out, n, o.h, o.w, o.cin, o.oh, o.ow, o.cout, o.kh, o.kw, o.stride, o.pad_t, o.pad_l, o.act, s,
                                              o.res_buf >= 0 ? (o.reserved >> 12) & 3 : 0, (o.reserved >> 14) & 0x1FF, (o.reserved >> 23) & 0x1FF);
                    break;
                }
                rc = launch_conv2d_f32((const float*)in, (const float*)blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off),
                                       (const float*)blob_ptr(e, o.shift_off), o.res_buf >= 0 ? (const float*)tab[o.res_buf] : nullptr,
                                       (float*)out, n, o.h, o.w, o.cin, o.oh, o.ow, o.cout, o.kh, o.kw, o.stride, o.pad_t, o.pad_l, o.act, s);
                break;
            case HSEFR_OP_MAXPOOL_F32:
                rc = launch_maxpool_f32((const float*)in, (float*)out, n, o.h, o.w, o.cin, o.oh, o.ow, o.kh, o.stride, o.pad_t, o.pad_l, s);
                break;
            case HSEFR_OP_STEM7X7_BF16:
                rc = launch_stem7x7_bf16((const float*)in, blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off),
                                         (const float*)blob_ptr(e, o.shift_off), out, n, o.h, o.w, o.oh, o.ow, o.act, s);
                break;
            case HSEFR_OP_STEM7X7_POOL_BF16:
                rc = launch_stem7x7_pool_bf16((const float*)in, blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off),
                                              (const float*)blob_ptr(e, o.shift_off), out, n, o.h, o.w, o.oh, o.ow, o.reserved & 15,
                                              (o.reserved >> 4) & 15, s, i < e->d_derived.size() ? e->d_derived[i] : nullptr);
                break;
            case HSEFR_OP_MAXPOOL_BF16:
                rc = launch_maxpool3x3s2_bf16(in, out, n, o.h, o.w, o.cin, o.oh, o.ow, o.pad_t, o.pad_l, s);
                break;
            case HSEFR_OP_GAP_BF16:
                rc = launch_gap_bf16(in, (float*)out, n, o.h * o.w, o.cin, s);
                break;
            case HSEFR_OP_DWPW_F32:
                rc = launch_dwpw_fused((const float*)in, (const float*)blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off),
                                       (const float*)blob_ptr(e, o.shift_off), (const float*)blob_ptr(e, o.w2_off),
                                       (const float*)blob_ptr(e, o.shift2_off), (float*)out, n, o.h, o.w, o.cin, o.stride,
                                       o.pad_t, o.pad_l, o.oh, o.ow, o.cout, HSEFR_ACT_RELU6, o.act, s);
                break;
            case HSEFR_OP_STEM3_F16S: {
                const float* pk = (const float*)blob_ptr(e, o.w_off);
                const float* ds2 = (const float*)blob_ptr(e, o.shift2_off);
                const int h1 = (o.h + 1) / 2, w1 = (o.w + 1) / 2;
                // pack: [0, 1952) the fp32 constants of stem2, [1952, 3008) conv1 split rows + descale for stem3_fused.hip,
                // [3008, 5056) conv1 in the two-step K layout of stem4_fused.hip, [5056, 7104) the same channel-reversed for uint8
                // RGB input, [7104, 7232) its four mean-folded shift vectors, [7232, 7264) its descale
                if (input_u8) {
                    rc = (g_stem5 ? launch_stem5_stream : launch_stem4_fused)(in, 1, pk + 5056, pk + 7232, pk + 7104, pk + 896, pk + 1184, pk + 1216,
                                                                            blob_ptr(e, o.w2_off), ds2, ds2 + 64, pk + 1248, pk + 1824, pk + 1888,
                                                                            (float*)out, nullptr, n, o.h, o.w, 0, o.reserved & 255, o.act, s);
                    break;
                }
                if (g_stem4 && stem4_route(o)) {
                    rc = (g_stem5 ? launch_stem5_stream : launch_stem4_fused)(in, 0, pk + 3008, pk + 1952 + 1024, pk + 864, pk + 896, pk + 1184, pk + 1216,
                                                                            blob_ptr(e, o.w2_off), ds2, ds2 + 64, pk + 1248, pk + 1824, pk + 1888,
                                                                            (float*)out, e->d_overflow, n, o.h, o.w, ((o.reserved >> 8) & 255) - 64,
                                                                            o.reserved & 255, o.act, s);
                    break;
                }
                rc = launch_stem3_fused((const float*)in, pk + 1952, pk + 1952 + 1024, pk + 864, pk + 896, pk + 1184, pk + 1216,
                                        blob_ptr(e, o.w2_off), ds2, ds2 + 64, pk + 1248, pk + 1824, pk + 1888, (float*)out, e->d_overflow, n,
                                        o.h, o.w, o.pad_t, o.pad_l, h1, w1, (o.kw >> 4) & 1, (o.kw >> 5) & 1, o.oh, o.ow,
                                        ((o.reserved >> 8) & 255) - 64, o.reserved & 255, o.act, s);
                break;
            }
            case HSEFR_OP_STEM2_F16S: {
                const float* pk = (const float*)blob_ptr(e, o.w_off);
                const float* ds2 = (const float*)blob_ptr(e, o.shift2_off);
                const int h1 = (o.h + 1) / 2, w1 = (o.w + 1) / 2;
                rc = launch_stem2_fused((const float*)in, pk, pk + 864, pk + 896, pk + 1184, pk + 1216, blob_ptr(e, o.w2_off), ds2,
                                        ds2 + 64, pk + 1248, pk + 1824, pk + 1888, (float*)out, n, o.h, o.w, o.pad_t, o.pad_l, h1, w1,
                                        (o.kw >> 4) & 1, (o.kw >> 5) & 1, o.oh, o.ow, o.reserved, o.act, s);
                break;
            }
#ifdef HSEFR_DEV
            case HSEFR_OP_STEM_F16S: {
                const float* pk = (const float*)blob_ptr(e, o.w_off);
                const float* ds2 = (const float*)blob_ptr(e, o.shift2_off);
                rc = launch_stem_fused((const float*)in, pk, pk + 864, pk + 896, pk + 1184, pk + 1216, blob_ptr(e, o.w2_off), ds2,
                                       ds2 + o.cout, (float*)out, n, o.h, o.w, o.pad_t, o.pad_l, o.oh, o.ow, o.reserved, o.act, s);
                break;
            }
#endif
            case HSEFR_OP_DWPW_F16S: {
                const float* ds2 = (const float*)blob_ptr(e, o.shift2_off);
                rc = launch_dwpw_f16s((const float*)in, (const float*)blob_ptr(e, o.w_off), (const float*)blob_ptr(e, o.scale_off),
                                      (const float*)blob_ptr(e, o.shift_off), blob_ptr(e, o.w2_off), ds2, ds2 + o.cout,
                                      (float*)out, n, o.h, o.w, o.cin, o.stride, o.pad_t, o.pad_l, o.oh, o.ow, o.cout,
                                      o.reserved, o.act, s);
                break;
            }
            default:
                set_error("forward: op %zu has unknown kind %u", i, o.kind);
                rc = HSEFR_ERR_UNSUPPORTED;
        }
        if (rc != HSEFR_OK) return rc;
        for (size_t k = 0; k <= covered; ++k)         // (the covered ops' own intervals are empty: their time is the flagged op's)
            if (prof) HSEFR_HIP_CHECK(hipEventRecord(pev[i + 1 + k], s));
        i += covered;
    }
    set_sweep_reverse(0);
    return HSEFR_OK;
}

static int engine_forward(hsefr_engine* e, const void* d_input, bool input_u8, int n, void* d_features, void* d_age_probs,
                          void* d_gender, hsefr_stream_t stream);

int hsefr_engine_forward(hsefr_engine* e, const void* d_input, int n, void* d_features, void* d_age_probs,
                         void* d_gender, hsefr_stream_t stream) {
    return engine_forward(e, d_input, false, n, d_features, d_age_probs, d_gender, stream);
}

int hsefr_engine_accepts_u8(const hsefr_engine* e) {
    if (!e || e->ops.empty()) return 0;
    const hsefr_plan_op& o = e->ops[0];
    return o.kind == HSEFR_OP_STEM3_F16S && o.in_buf == HSEFR_BUF_INPUT && ((o.reserved >> 16) & 1) && stem4_route(o);
}

int hsefr_engine_forward_u8(hsefr_engine* e, const void* d_input_u8, int n, void* d_features, void* d_age_probs,
                            void* d_gender, hsefr_stream_t stream) {
    HSEFR_REQUIRE(e, HSEFR_ERR_INVALID, "forward_u8: null engine");
    HSEFR_REQUIRE(hsefr_engine_accepts_u8(e), HSEFR_ERR_UNSUPPORTED,
                  "forward_u8: this plan takes no uint8 input (it needs the fused stem lowered with a BGR mean and an input whose "
                  "edges are multiples of 4)");
    return engine_forward(e, d_input_u8, true, n, d_features, d_age_probs, d_gender, stream);
}

static int engine_forward(hsefr_engine* e, const void* d_input, bool input_u8, int n, void* d_features, void* d_age_probs,
                          void* d_gender, hsefr_stream_t stream) {
    HSEFR_REQUIRE(e, HSEFR_ERR_INVALID, "forward: null engine");
    HSEFR_REQUIRE(d_input || n == 0, HSEFR_ERR_INVALID, "forward: null input");
    HSEFR_REQUIRE(n >= 0 && n <= e->max_batch, HSEFR_ERR_SHAPE, "forward: batch %d outside [0, %d]", n, e->max_batch);
    void* outs[HSEFR_N_OUTPUT_SLOTS] = {d_features, d_age_probs, d_gender};
    for (int s = 0; s < HSEFR_N_OUTPUT_SLOTS; ++s)
        HSEFR_REQUIRE(!outs[s] || e->hdr.out_buffer[s] != HSEFR_BUF_NONE, HSEFR_ERR_INVALID,
                      "forward: the plan does not produce output slot %d", s);
    if (n == 0) return HSEFR_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool prof = e->prof_depth > 0;
    hipEvent_t* pev = prof ? e->events.data() + (size_t)(e->prof_calls % e->prof_depth) * (e->ops.size() + 1) : nullptr;
    // Like sess.run, evaluate only what the requested fetches need: walk the op list backwards
    // from the requested output buffers (buffers are reused, so liveness is positional).
    // With no output pointer at all, every op runs (per-layer parity tests read the buffers).
    std::vector<char> needed(e->ops.size(), 1);
    if (d_features || d_age_probs || d_gender) {
        std::vector<char> live(e->d_bufs.size(), 0);
        for (int sl = 0; sl < HSEFR_N_OUTPUT_SLOTS; ++sl)
            if (outs[sl]) live[e->hdr.out_buffer[sl]] = 1;
        for (size_t i = e->ops.size(); i-- > 0;) {
            const hsefr_plan_op& o = e->ops[i];
            needed[i] = live[o.out_buf];
            if (!needed[i]) continue;
            live[o.out_buf] = 0;
            if (o.in_buf >= 0) live[o.in_buf] = 1;
            if (o.res_buf >= 0) live[o.res_buf] = 1;
        }
    }
    // ---- small batches: replay a captured graph (see hsefr_engine::graph_max_n) ----
    const int mask = (d_features ? 1 : 0) | (d_age_probs ? 2 : 0) | (d_gender ? 4 : 0);
    if (!prof && !input_u8 && mask != 0 && n <= e->graph_max_n && e->ops.size() > 0 && e->ops[0].in_buf == HSEFR_BUF_INPUT) {
        const size_t in_bytes = (size_t)e->hdr.in_h * e->hdr.in_w * e->hdr.in_c * sizeof(float);
        if (!e->d_in_stage) {
            HSEFR_HIP_CHECK(hipMalloc(&e->d_in_stage, in_bytes * e->graph_max_n));
            e->device_bytes += in_bytes * e->graph_max_n;
            HSEFR_HIP_CHECK(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
        }
        hipGraphExec_t exec = nullptr;
        for (const auto& g : e->graphs)
            if (g.n == n && g.mask == mask) exec = g.exec;
        if (!exec) {
            hipGraph_t graph = nullptr;
            HSEFR_HIP_CHECK(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
            const int rc = run_ops(e, e->d_bufs, e->d_in_stage, n, needed, e->cap_stream, nullptr);   // captured pointers must not change
            const hipError_t ce = hipStreamEndCapture(e->cap_stream, &graph);
            if (rc != HSEFR_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            HSEFR_HIP_CHECK(ce);
            HSEFR_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
            e->graphs.push_back({n, mask, exec});
        }
        HSEFR_HIP_CHECK(hipMemcpyAsync(e->d_in_stage, d_input, in_bytes * n, hipMemcpyDeviceToDevice, s));
        HSEFR_HIP_CHECK(hipGraphLaunch(exec, s));
        e->graph_launches++;
        for (int sl = 0; sl < HSEFR_N_OUTPUT_SLOTS; ++sl) {      // a replayed graph writes the engine's own buffers
            if (!outs[sl]) continue;
            const size_t bytes = (size_t)e->hdr.out_elems[sl] * sizeof(float) * n;
            HSEFR_HIP_CHECK(hipMemcpyAsync(outs[sl], e->d_bufs[e->hdr.out_buffer[sl]], bytes, hipMemcpyDeviceToDevice, s));
        }
    } else {
        // Output buffers are pinned by the lowering (one producer, never recycled, distinct per slot), so the caller's
        // pointer can stand in for the whole forward.  A buffer serving two slots is written once and copied once.
        // The kernels store 16 bytes per lane: a caller pointer that is not 16-byte aligned (the ABI does not require it) keeps
        // the engine's own buffer for that slot and is served by a copy, like a slot that shares its buffer with another.
        std::vector<void*> tab(e->d_bufs);
        const void* src_of[HSEFR_N_OUTPUT_SLOTS] = {nullptr, nullptr, nullptr};
        for (int sl = 0; sl < HSEFR_N_OUTPUT_SLOTS; ++sl) {
            if (!outs[sl]) continue;
            const int b = e->hdr.out_buffer[sl];
            if (tab[b] != e->d_bufs[b]) { src_of[sl] = tab[b]; continue; }             // an earlier slot already redirected this buffer
            if (((uintptr_t)outs[sl] & 15) == 0) tab[b] = outs[sl];                  // written in place by its producer
            else src_of[sl] = e->d_bufs[b];
        }
        for (int sl = 0; sl < HSEFR_N_OUTPUT_SLOTS; ++sl)                             // an unaligned slot ahead of an aligned one of the same buffer
            if (outs[sl] && src_of[sl] == e->d_bufs[e->hdr.out_buffer[sl]]) src_of[sl] = tab[e->hdr.out_buffer[sl]];
        const int rc = run_ops(e, tab, d_input, n, needed, s, prof ? pev : nullptr, input_u8);
        if (rc != HSEFR_OK) return rc;
        for (int sl = 0; sl < HSEFR_N_OUTPUT_SLOTS; ++sl)
            if (outs[sl] && src_of[sl])
                HSEFR_HIP_CHECK(hipMemcpyAsync(outs[sl], src_of[sl], (size_t)e->hdr.out_elems[sl] * sizeof(float) * n,
                                               hipMemcpyDeviceToDevice, s));
    }
    if (prof) e->prof_calls++;
    return HSEFR_OK;
}

int hsefr_plan_describe(const void* plan, size_t plan_bytes, int n, char* out, size_t out_bytes) {
    HSEFR_REQUIRE(out && out_bytes > 0 && n > 0, HSEFR_ERR_INVALID, "plan_describe: bad argument");
    out[0] = 0;
    hsefr_engine e;                      // tables only: no device memory, no HIP call (the probe keeps every launcher off the device)
    const char* blob = nullptr;
    int rc = check_plan_blob(plan, plan_bytes, e.hdr, e.bufs, e.ops, blob);
    if (rc != HSEFR_OK) return rc;
    e.max_batch = n;
    // stand-in addresses (never dereferenced: nothing is launched): distinct, 16-byte aligned, non-null
    e.d_blob = reinterpret_cast<char*>(uintptr_t(1) << 40);
    e.d_overflow = reinterpret_cast<int*>(uintptr_t(1) << 39);
    e.d_bufs.resize(e.hdr.n_buffers);
    for (uint32_t i = 0; i < e.hdr.n_buffers; ++i) e.d_bufs[i] = reinterpret_cast<void*>((uintptr_t(2) << 40) + (uintptr_t(i) << 32));
    std::string text;
    std::vector<char> only(e.ops.size(), 0);
    for (size_t i = 0; i < e.ops.size(); ++i) {
        // op i alone, then with the ops its flags cover (a flagged op's launch needs them `needed`)
        std::fill(only.begin(), only.end(), 0);
        only[i] = 1;
        const size_t span = (e.ops[i].flags & HSEFR_OPF_PAIR_NEXT) ? 1 : (e.ops[i].flags & HSEFR_OPF_HEADS) ? 3 : 0;
        for (size_t k = 1; k <= span && i + k < e.ops.size(); ++k) only[i + k] = 1;
        std::string line = std::to_string(i) + "\t" + std::to_string(e.ops[i].kind) + "\t";
        g_route_sink = &line;
        rc = run_ops(&e, e.d_bufs, reinterpret_cast<const void*>(uintptr_t(3) << 40), n, only, nullptr, nullptr,
                     false);
        g_route_sink = nullptr;
        if (rc != HSEFR_OK) { e.d_blob = nullptr; e.d_overflow = nullptr; e.d_bufs.clear(); return rc; }
        text += line + "\n";
        for (size_t k = 1; k <= span && i + k < e.ops.size(); ++k)
            text += std::to_string(i + k) + "\t" + std::to_string(e.ops[i + k].kind) + "\t(inside op " + std::to_string(i) + ")\n";
        i += span;
    }
    e.d_blob = nullptr; e.d_overflow = nullptr; e.d_bufs.clear();     // (stand-ins: nothing to free)
    HSEFR_REQUIRE(text.size() + 1 <= out_bytes, HSEFR_ERR_INVALID, "plan_describe: the table needs %zu bytes, the buffer holds %zu", text.size() + 1, out_bytes);
    memcpy(out, text.c_str(), text.size() + 1);
    return HSEFR_OK;
}

int hsefr_engine_input_overflow(hsefr_engine* e, int* host_flag, hsefr_stream_t stream) {
    HSEFR_REQUIRE(e && host_flag, HSEFR_ERR_INVALID, "input_overflow: null argument");
    HSEFR_HIP_CHECK(hipMemcpyAsync(host_flag, e->d_overflow, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HSEFR_HIP_CHECK(hipMemsetAsync(e->d_overflow, 0, sizeof(int), (hipStream_t)stream));
    HSEFR_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return HSEFR_OK;
}

int hsefr_engine_input_overflow_async(hsefr_engine* e, int* pinned_host_flag, hsefr_stream_t stream) {
    HSEFR_REQUIRE(e && pinned_host_flag, HSEFR_ERR_INVALID, "input_overflow_async: null argument");
    // enqueue only: the copy and the clear are ordered on `stream` behind the forwards already queued there; the caller waits on
    // an event of its own before it reads the (page-locked) int
    HSEFR_HIP_CHECK(hipMemcpyAsync(pinned_host_flag, e->d_overflow, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HSEFR_HIP_CHECK(hipMemsetAsync(e->d_overflow, 0, sizeof(int), (hipStream_t)stream));
    return HSEFR_OK;
}

int hsefr_engine_set_graph_batch(hsefr_engine* e, int max_n) {
    HSEFR_REQUIRE(e && max_n >= 0 && max_n <= e->max_batch, HSEFR_ERR_INVALID, "set_graph_batch: bad argument");
    for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.exec);
    e->graphs.clear();
    if (e->d_in_stage) {
        (void)hipFree(e->d_in_stage);
        e->d_in_stage = nullptr;
        e->device_bytes -= (size_t)e->hdr.in_h * e->hdr.in_w * e->hdr.in_c * sizeof(float) * e->graph_max_n;
        (void)hipStreamDestroy(e->cap_stream);
        e->cap_stream = nullptr;
    }
    e->graph_max_n = max_n;
    return HSEFR_OK;
}

long long hsefr_engine_graph_launches(const hsefr_engine* e) { return e ? e->graph_launches : 0; }

int hsefr_engine_destroy(hsefr_engine* e) {
    if (!e) return HSEFR_OK;
    for (void* b : e->d_bufs)
        if (b) (void)hipFree(b);
    if (e->d_blob) (void)hipFree(e->d_blob);
    if (e->d_overflow) (void)hipFree(e->d_overflow);
    for (void* b : e->d_derived)
        if (b) (void)hipFree(b);
    for (auto ev : e->events) (void)hipEventDestroy(ev);
    for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.exec);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    if (e->d_in_stage) (void)hipFree(e->d_in_stage);
    delete e;
    return HSEFR_OK;
}

// ---- per-kernel entry points ---------------------------------------------------------------
int hsefr_conv_c3_bias_act(const float* x, const float* wgt, const float* shift, float* y, int n, int h, int w,
                           int kh, int kw, int stride, int pad_t, int pad_l, int oh, int ow, int cout, int act,
                           hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wgt && shift && y), HSEFR_ERR_INVALID, "conv_c3: null pointer");
    return launch_conv_c3(x, wgt, shift, y, n, h, w, kh, kw, stride, pad_t, pad_l, oh, ow, cout, act, (hipStream_t)stream);
}

int hsefr_dwconv3x3_bn_relu6(const float* x, const float* wgt, const float* scale, const float* shift, float* y,
                             int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh, int ow, int act,
                             hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wgt && scale && shift && y), HSEFR_ERR_INVALID, "dwconv3x3: null pointer");
    return launch_dwconv3x3(x, wgt, scale, shift, y, n, h, w, c, stride, pad_t, pad_l, oh, ow, act, (hipStream_t)stream);
}

int hsefr_dwconv3x3_bn_relu6_split(const float* x, const float* wgt, const float* scale, const float* shift, void* y_split,
                                   int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh, int ow, int act,
                                   int a_log2, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wgt && scale && shift && y_split), HSEFR_ERR_INVALID, "dwconv3x3_split: null pointer");
    return launch_dwconv3x3_split(x, wgt, scale, shift, y_split, n, h, w, c, stride, pad_t, pad_l, oh, ow, act, a_log2, (hipStream_t)stream);
}

int hsefr_pwconv1x1_presplit_gap(const void* x_split, const void* w_split, const float* descale, const float* shift, float* y,
                                 long long m, int k, int cout, int act, int map_hw, hsefr_stream_t stream) {
    HSEFR_REQUIRE(m == 0 || (x_split && w_split && descale && shift && y), HSEFR_ERR_INVALID, "pwconv1x1_presplit_gap: null pointer");
    return launch_pwconv_ps_gap(x_split, w_split, descale, shift, y, m, k, cout, act, map_hw, (hipStream_t)stream);
}

int hsefr_pwconv1x1_presplit_dw(const void* x_split, const void* w_split, const float* descale, const float* shift, const float* dw_consts,
                                void* y_split, long long m, int k, int cout, int act, int map_w, int map_hw, int dw_stride, int out_log2,
                                hsefr_stream_t stream) {
    HSEFR_REQUIRE(m == 0 || (x_split && w_split && descale && shift && dw_consts && y_split), HSEFR_ERR_INVALID, "pwconv1x1_presplit_dw: null pointer");
    return launch_pwconv_ps_dw(x_split, w_split, descale, shift, dw_consts, y_split, m, k, cout, act, map_w, map_hw, dw_stride, out_log2,
                               (hipStream_t)stream);
}

int hsefr_pwconv1x1_presplit(const void* x_split, const void* w_split, const float* descale, const float* shift, float* y,
                             long long m, int k, int cout, int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(m == 0 || (x_split && w_split && descale && shift && y), HSEFR_ERR_INVALID, "pwconv_presplit: null pointer");
    return launch_pwconv_ps(x_split, w_split, descale, shift, y, m, k, cout, act, (hipStream_t)stream);
}

int hsefr_pwconv1x1_bias_relu6(const float* x, const float* wgt_t, const float* shift, float* y, long long m, int k,
                               int cout, int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(m == 0 || (x && wgt_t && shift && y), HSEFR_ERR_INVALID, "pwconv: null pointer");
    return launch_pwconv_f32(x, wgt_t, shift, y, m, k, cout, act, (hipStream_t)stream);
}

int hsefr_pwconv1x1_f16split(const float* x, const void* w_split, const float* descale, const float* shift, float* y,
                             long long m, int k, int cout, int a_log2, int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(m == 0 || (x && w_split && descale && shift && y), HSEFR_ERR_INVALID, "pwconv_f16split: null pointer");
    return launch_pwconv_f16s(x, w_split, descale, shift, y, m, k, cout, a_log2, act, (hipStream_t)stream);
}

int hsefr_stem2_fused(const float* x, const float* conv_w, const float* conv_shift, const float* wd1, const float* d1scale,
                      const float* d1shift, const void* w_split, const float* descale, const float* pshift, const float* wd2,
                      const float* d2scale, const float* d2shift, float* y, int n, int h, int w, int cpad_t, int cpad_l,
                      int h1, int w1, int pad_t2, int pad_l2, int oh2, int ow2, int a_log2, int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && conv_w && conv_shift && wd1 && d1scale && d1shift && w_split && descale && pshift && wd2 &&
                             d2scale && d2shift && y), HSEFR_ERR_INVALID, "stem2_fused: null pointer");
    return launch_stem2_fused(x, conv_w, conv_shift, wd1, d1scale, d1shift, w_split, descale, pshift, wd2, d2scale, d2shift, y, n,
                              h, w, cpad_t, cpad_l, h1, w1, pad_t2, pad_l2, oh2, ow2, a_log2, act, (hipStream_t)stream);
}

int hsefr_stem3_fused(const float* x, const void* cw_split, const float* cdescale, const float* conv_shift, const float* wd1,
                      const float* d1scale, const float* d1shift, const void* w_split, const float* descale, const float* pshift,
                      const float* wd2, const float* d2scale, const float* d2shift, float* y, int* d_overflow, int n, int h, int w,
                      int cpad_t, int cpad_l, int h1, int w1, int pad_t2, int pad_l2, int oh2, int ow2, int in_log2, int a_log2,
                      int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && cw_split && cdescale && conv_shift && wd1 && d1scale && d1shift && w_split && descale && pshift && wd2 &&
                             d2scale && d2shift && y), HSEFR_ERR_INVALID, "stem3_fused: null pointer");
    return launch_stem3_fused(x, cw_split, cdescale, conv_shift, wd1, d1scale, d1shift, w_split, descale, pshift, wd2, d2scale, d2shift, y,
                              d_overflow, n, h, w, cpad_t, cpad_l, h1, w1, pad_t2, pad_l2, oh2, ow2, in_log2, a_log2, act, (hipStream_t)stream);
}

int hsefr_stem4_fused(const void* x, int x_is_u8, const void* cw4, const float* cdescale, const float* conv_shift, const float* wd1,
                      const float* d1scale, const float* d1shift, const void* w_split, const float* descale, const float* pshift,
                      const float* wd2, const float* d2scale, const float* d2shift, float* y, int* d_overflow, int n, int h, int w,
                      int in_log2, int a_log2, int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && cw4 && cdescale && conv_shift && wd1 && d1scale && d1shift && w_split && descale && pshift && wd2 &&
                             d2scale && d2shift && y), HSEFR_ERR_INVALID, "stem4_fused: null pointer");
    return launch_stem4_fused(x, x_is_u8, cw4, cdescale, conv_shift, wd1, d1scale, d1shift, w_split, descale, pshift, wd2, d2scale, d2shift, y,
                              d_overflow, n, h, w, in_log2, a_log2, act, (hipStream_t)stream);
}

int hsefr_stem5_stream(const void* x, int x_is_u8, const void* cw4, const float* cdescale, const float* conv_shift, const float* wd1,
                       const float* d1scale, const float* d1shift, const void* w_split, const float* descale, const float* pshift,
                       const float* wd2, const float* d2scale, const float* d2shift, float* y, int* d_overflow, int n, int h, int w,
                       int in_log2, int a_log2, int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && cw4 && cdescale && conv_shift && wd1 && d1scale && d1shift && w_split && descale && pshift && wd2 &&
                             d2scale && d2shift && y), HSEFR_ERR_INVALID, "stem5_stream: null pointer");
    return launch_stem5_stream(x, x_is_u8, cw4, cdescale, conv_shift, wd1, d1scale, d1shift, w_split, descale, pshift, wd2, d2scale, d2shift, y,
                               d_overflow, n, h, w, in_log2, a_log2, act, (hipStream_t)stream);
}

#ifdef HSEFR_DEV
int hsefr_stem_fused(const float* x, const float* conv_w, const float* conv_shift, const float* wd, const float* dscale,
                     const float* dshift, const void* w_split, const float* descale, const float* pshift, float* y, int n,
                     int h, int w, int cpad_t, int cpad_l, int oh, int ow, int a_log2, int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && conv_w && conv_shift && wd && dscale && dshift && w_split && descale && pshift && y),
                  HSEFR_ERR_INVALID, "stem_fused: null pointer");
    return launch_stem_fused(x, conv_w, conv_shift, wd, dscale, dshift, w_split, descale, pshift, y, n, h, w, cpad_t, cpad_l,
                             oh, ow, a_log2, act, (hipStream_t)stream);
}
#endif

int hsefr_dwpw_f16split(const float* x, const float* wd, const float* dscale, const float* dshift, const void* w_split,
                        const float* descale, const float* pshift, float* y, int n, int h, int w, int c, int stride,
                        int pad_t, int pad_l, int oh, int ow, int cout, int a_log2, int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wd && dscale && dshift && w_split && descale && pshift && y), HSEFR_ERR_INVALID,
                  "dwpw_f16split: null pointer");
    return launch_dwpw_f16s(x, wd, dscale, dshift, w_split, descale, pshift, y, n, h, w, c, stride, pad_t, pad_l, oh, ow, cout,
                            a_log2, act, (hipStream_t)stream);
}

int hsefr_dwpw_fused(const float* x, const float* wd, const float* dscale, const float* dshift, const float* wp_t,
                     const float* pshift, float* y, int n, int h, int w, int c, int stride, int pad_t, int pad_l, int oh,
                     int ow, int cout, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wd && dscale && dshift && wp_t && pshift && y), HSEFR_ERR_INVALID, "dwpw_fused: null pointer");
    return launch_dwpw_fused(x, wd, dscale, dshift, wp_t, pshift, y, n, h, w, c, stride, pad_t, pad_l, oh, ow, cout,
                             HSEFR_ACT_RELU6, HSEFR_ACT_RELU6, (hipStream_t)stream);
}

int hsefr_gap(const float* x, float* y, int n, int hw, int c, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && y), HSEFR_ERR_INVALID, "gap: null pointer");
    return launch_gap(x, y, n, hw, c, (hipStream_t)stream);
}

int hsefr_dense(const float* x, const float* wgt, const float* bias, float* y, int n, int k, int cout, int act,
                hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wgt && y), HSEFR_ERR_INVALID, "dense: null pointer");
    return launch_dense(x, wgt, bias, y, n, k, cout, act, (hipStream_t)stream);
}

int hsefr_heads_fused(const float* x, const float* w1, const float* b1, const float* wa, const float* ba, const float* wg, const float* bg,
                      float* hidden, float* logits, float* age_probs, float* gender, int n, int k, int a, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && w1 && b1 && wa && ba && wg && bg && hidden && logits && age_probs && gender), HSEFR_ERR_INVALID,
                  "heads_fused: null pointer");
    return launch_heads_fused(x, w1, b1, wa, ba, wg, bg, hidden, logits, age_probs, gender, n, k, a, (hipStream_t)stream);
}

int hsefr_softmax(const float* x, float* y, int n, int c, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && y), HSEFR_ERR_INVALID, "softmax: null pointer");
    return launch_softmax(x, y, n, c, (hipStream_t)stream);
}

int hsefr_conv_bf16(const void* x, const void* wgt_t, const float* scale, const float* shift, const void* res, void* y,
                    int n, int h, int w, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t,
                    int pad_l, int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wgt_t && scale && shift && y), HSEFR_ERR_INVALID, "conv_bf16: null pointer");
    return launch_conv_bf16(x, wgt_t, scale, shift, res, y, n, h, w, c, oh, ow, cout, kh, kw, stride, pad_t, pad_l, act,
                            (hipStream_t)stream);
}

int hsefr_conv1x1_proj_bf16(const void* x, const void* wgt_t, const float* scale, const float* shift, const void* x2, const void* wgt2_t,
                            const float* scale2, const float* shift2, void* y, int n, int oh, int ow, int c, int cout, int c2, int stride2,
                            int h2, int w2, int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wgt_t && scale && shift && x2 && wgt2_t && scale2 && shift2 && y), HSEFR_ERR_INVALID, "conv1x1_proj_bf16: null pointer");
    return launch_conv1x1_proj_bf16(x, wgt_t, scale, shift, x2, wgt2_t, scale2, shift2, y, n, oh, ow, c, cout, c2, stride2, h2, w2, act,
                                    (hipStream_t)stream);
}

int hsefr_conv1x1_sres_bf16(const void* x, const void* wgt_t, const float* scale, const float* shift, const void* res, void* y, int n, int oh,
                            int ow, int c, int cout, int res_stride, int h2, int w2, int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wgt_t && scale && shift && res && y), HSEFR_ERR_INVALID, "conv1x1_sres_bf16: null pointer");
    return launch_conv1x1_sres_bf16(x, wgt_t, scale, shift, res, y, n, oh, ow, c, cout, res_stride, h2, w2, act, (hipStream_t)stream);
}

int hsefr_conv1x1_pair_bf16(const void* x, const void* w1_t, const float* scale1, const float* shift1, const void* res, const void* x2,
                            const void* wp_t, const float* scale_p, const float* shift_p, void* y1, const void* w2_t, const float* scale2,
                            const float* shift2, void* y2, long long pixels, int c, int cout1, int cout2, int c2, int act1, int act2,
                            hsefr_stream_t stream) {
    HSEFR_REQUIRE(pixels >= 0, HSEFR_ERR_INVALID, "conv1x1_pair_bf16: pixels=%lld", pixels);
    if (pixels == 0) return HSEFR_OK;
    HSEFR_REQUIRE(x && w1_t && scale1 && shift1 && y1 && w2_t && scale2 && shift2 && y2 && (c2 == 0 ? res != nullptr : (x2 && wp_t && scale_p && shift_p)),
                  HSEFR_ERR_INVALID, "conv1x1_pair_bf16: null pointer");
    return launch_conv1x1_pair_bf16(x, w1_t, scale1, shift1, c2 == 0 ? res : nullptr, x2, wp_t, scale_p, shift_p, y1, w2_t, scale2, shift2, y2, pixels,
                                    c, cout1, cout2, c2, act1, act2, (hipStream_t)stream);
}

int hsefr_conv1x1_pair_sub2_bf16(const void* x, const void* w1_t, const float* scale1, const float* shift1, const void* res, const void* x2,
                                 const void* wp_t, const float* scale_p, const float* shift_p, void* y1, const void* w2_t, const float* scale2,
                                 const float* shift2, void* y2, int n, int h, int w, int c, int cout1, int cout2, int c2, int act1, int act2,
                                 hsefr_stream_t stream) {
    HSEFR_REQUIRE(n >= 0 && h > 1 && w > 1, HSEFR_ERR_INVALID, "conv1x1_pair_sub2_bf16: %d maps of %dx%d", n, h, w);
    if (n == 0) return HSEFR_OK;
    HSEFR_REQUIRE(x && w1_t && scale1 && shift1 && y1 && w2_t && scale2 && shift2 && y2 && (c2 == 0 ? res != nullptr : (x2 && wp_t && scale_p && shift_p)),
                  HSEFR_ERR_INVALID, "conv1x1_pair_sub2_bf16: null pointer");
    return launch_conv1x1_pair_bf16(x, w1_t, scale1, shift1, c2 == 0 ? res : nullptr, x2, wp_t, scale_p, shift_p, y1, w2_t, scale2, shift2, y2,
                                    (long long)n * h * w, c, cout1, cout2, c2, act1, act2, (hipStream_t)stream, 2, h, w);
}

int hsefr_stem7x7_bf16(const float* x, const void* wgt_t, const float* scale, const float* shift, void* y, int n, int h,
                       int w, int oh, int ow, int act, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wgt_t && scale && shift && y), HSEFR_ERR_INVALID, "stem7x7: null pointer");
    return launch_stem7x7_bf16(x, wgt_t, scale, shift, y, n, h, w, oh, ow, act, (hipStream_t)stream);
}

int hsefr_stem7x7_pool_bf16(const float* x, const void* wgt_t, const float* scale, const float* shift, void* y, int n, int h, int w,
                            int ph, int pw, int pool_pad_t, int pool_pad_l, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wgt_t && scale && shift && y), HSEFR_ERR_INVALID, "stem7x7_pool: null pointer");
    return launch_stem7x7_pool_bf16(x, wgt_t, scale, shift, y, n, h, w, ph, pw, pool_pad_t, pool_pad_l, (hipStream_t)stream);
}

int hsefr_maxpool3x3s2_bf16(const void* x, void* y, int n, int h, int w, int c, int oh, int ow, int pad_t, int pad_l,
                            hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && y), HSEFR_ERR_INVALID, "maxpool: null pointer");
    return launch_maxpool3x3s2_bf16(x, y, n, h, w, c, oh, ow, pad_t, pad_l, (hipStream_t)stream);
}

int hsefr_gap_bf16(const void* x, float* y, int n, int hw, int c, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && y), HSEFR_ERR_INVALID, "gap_bf16: null pointer");
    return launch_gap_bf16(x, y, n, hw, c, (hipStream_t)stream);
}

int hsefr_preprocess_pil_u8(const unsigned char* d_in, unsigned char* d_tmp, float* d_out, int n, int H, int W, int oh, int ow,
                            const int* d_xmin, const int* d_xcnt, const int* d_xcoef, int xksize, const int* d_ymin,
                            const int* d_ycnt, const int* d_ycoef, int yksize, int color_mode, const double* mean3,
                            hsefr_stream_t stream) {
    HSEFR_REQUIRE(mean3 && (n == 0 || (d_in && d_tmp && d_out && d_xmin && d_xcnt && d_xcoef && d_ymin && d_ycnt && d_ycoef)),
                  HSEFR_ERR_INVALID, "preprocess_pil: null pointer");
    return launch_pil_resize(d_in, d_tmp, d_out, n, H, W, oh, ow, d_xmin, d_xcnt, d_xcoef, xksize, d_ymin, d_ycnt, d_ycoef, yksize,
                             color_mode, mean3, (hipStream_t)stream);
}

int hsefr_preprocess_cv_u8(const unsigned char* d_in, float* d_out, int n, int H, int W, int oh, int ow, const int* d_x0,
                           const int* d_x1, const int* d_wx1, const int* d_y0, const int* d_y1, const int* d_wy1,
                           int color_mode, const double* mean3, hsefr_stream_t stream) {
    HSEFR_REQUIRE(mean3 && (n == 0 || (d_in && d_out)), HSEFR_ERR_INVALID, "preprocess_cv: null pointer");
    HSEFR_REQUIRE((H == oh && W == ow) || (d_x0 && d_x1 && d_wx1 && d_y0 && d_y1 && d_wy1), HSEFR_ERR_INVALID,
                  "preprocess_cv: null tap table");
    return launch_cv_resize(d_in, d_out, n, H, W, oh, ow, d_x0, d_x1, d_wx1, d_y0, d_y1, d_wy1, color_mode, mean3,
                            (hipStream_t)stream);
}

int hsefr_l2_normalize(const float* x, float* y, int n, int d, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && y), HSEFR_ERR_INVALID, "l2_normalize: null pointer");
    return launch_l2_normalize(x, y, n, d, (hipStream_t)stream);
}

int hsefr_nn1(const float* q, const float* g, int nq, int ng, int d, int* nn_index, float* nn_dist2,
              hsefr_stream_t stream) {
    HSEFR_REQUIRE(nq == 0 || (q && g && nn_index), HSEFR_ERR_INVALID, "nn1: null pointer");
    return launch_nn1(q, g, nq, ng, d, nn_index, nn_dist2, (hipStream_t)stream);
}

long long hsefr_nn1_fallbacks(void) { return nn1_fallbacks(); }

int hsefr_conv2d_direct(const float* x, const float* wgt, const float* bias, const float* alpha, float* y, int n, int h, int w, int c,
                        int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wgt && y), HSEFR_ERR_INVALID, "conv2d_direct: null pointer");
    return launch_conv2d_direct(x, wgt, bias, alpha, y, n, h, w, c, oh, ow, cout, kh, kw, stride, pad_t, pad_l, (hipStream_t)stream);
}

int hsefr_conv2d_f32_mfma(const float* x, const float* wgt, const float* scale, const float* shift, const float* res, float* y, int n, int h,
                          int w, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, int act,
                          hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wgt && y), HSEFR_ERR_INVALID, "conv2d_f32_mfma: null pointer");
    return launch_conv_f32_mfma(x, wgt, scale, shift, res, y, n, h, w, c, oh, ow, cout, kh, kw, stride, pad_t, pad_l, act, (hipStream_t)stream);
}

int hsefr_conv2d_f32(const float* x, const float* wgt, const float* scale, const float* shift, const float* res, float* y, int n, int h,
                     int w, int c, int oh, int ow, int cout, int kh, int kw, int stride, int pad_t, int pad_l, int act,
                     hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && wgt && y), HSEFR_ERR_INVALID, "conv2d_f32: null pointer");
    return launch_conv2d_f32(x, wgt, scale, shift, res, y, n, h, w, c, oh, ow, cout, kh, kw, stride, pad_t, pad_l, act, (hipStream_t)stream);
}

int hsefr_maxpool_f32(const float* x, float* y, int n, int h, int w, int c, int oh, int ow, int k, int stride, int pad_t, int pad_l,
                      hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (x && y), HSEFR_ERR_INVALID, "maxpool_f32: null pointer");
    return launch_maxpool_f32(x, y, n, h, w, c, oh, ow, k, stride, pad_t, pad_l, (hipStream_t)stream);
}

int hsefr_mtcnn_pyramid_level(const unsigned char* d_frame, float* d_dst, int sh, int sw, int dh, int dw, hsefr_stream_t stream) {
    HSEFR_REQUIRE(d_frame && d_dst, HSEFR_ERR_INVALID, "mtcnn_pyramid_level: null pointer");
    return launch_area_level(d_frame, d_dst, sh, sw, dh, dw, (hipStream_t)stream);
}

int hsefr_mtcnn_post_capacity(void) { return mtcnn_post_capacity(); }

int hsefr_mtcnn_stage1_level(const float* prob, const float* reg, int w, int h, double scale, float thr, double* found, int* counters,
                             hsefr_stream_t stream) {
    HSEFR_REQUIRE(prob && reg && found && counters, HSEFR_ERR_INVALID, "mtcnn_stage1_level: null pointer");
    return launch_mtcnn_stage1_level(prob, reg, w, h, scale, thr, found, counters, (hipStream_t)stream);
}

int hsefr_mtcnn_stage1_finish(const double* found, int* counters, double* boxes, int* crop_table, int img_w, int img_h,
                              hsefr_stream_t stream) {
    HSEFR_REQUIRE(found && counters && boxes && crop_table && img_w > 0 && img_h > 0, HSEFR_ERR_INVALID, "mtcnn_stage1_finish: bad argument");
    return launch_mtcnn_stage1_finish(found, counters, boxes, crop_table, img_w, img_h, (hipStream_t)stream);
}

int hsefr_mtcnn_stage_finish(int stage, const double* boxes_in, int n, const float* prob, const float* reg, const float* pts, float thr,
                             double* boxes_out, int* crop_table, float* points_out, int* counters, int img_w, int img_h,
                             hsefr_stream_t stream) {
    HSEFR_REQUIRE(counters && boxes_out && (n == 0 || (boxes_in && prob && reg)), HSEFR_ERR_INVALID, "mtcnn_stage_finish: null pointer");
    HSEFR_REQUIRE(stage == 2 ? crop_table != nullptr : (stage == 3 && (n == 0 || pts) && points_out), HSEFR_ERR_INVALID,
                  "mtcnn_stage_finish: stage %d needs %s", stage, stage == 2 ? "a crop table" : "landmarks in and out");
    return launch_mtcnn_stage23_finish(stage, boxes_in, n, prob, reg, pts, thr, boxes_out, crop_table, points_out, counters, img_w, img_h,
                                       (hipStream_t)stream);
}

int hsefr_mtcnn_nms(const double* boxes, int n, double thr, int use_min, int* keep, int* n_keep, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n_keep && (n == 0 || (boxes && keep)), HSEFR_ERR_INVALID, "mtcnn_nms: null pointer");
    return launch_mtcnn_nms(boxes, n, thr, use_min, keep, n_keep, (hipStream_t)stream);
}

int hsefr_mtcnn_crops(const unsigned char* d_frame, const int* d_boxes, float* d_dst, int sh, int sw, int n, int size,
                      hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || (d_frame && d_boxes && d_dst), HSEFR_ERR_INVALID, "mtcnn_crops: null pointer");
    return launch_area_crops(d_frame, d_boxes, d_dst, sh, sw, n, size, (hipStream_t)stream);
}

int hsefr_pairwise_dist(const float* x, const float* y, int n, int m, int d, float* out, hsefr_stream_t stream) {
    HSEFR_REQUIRE(n == 0 || m == 0 || (x && y && out), HSEFR_ERR_INVALID, "pairwise_dist: null pointer");
    return launch_pairwise_dist(x, y, n, m, d, out, (hipStream_t)stream);
}

}  // extern "C"
#pragma GCC visibility pop
