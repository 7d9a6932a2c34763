"""Per-kernel Python entry points: torch CUDA tensors in, torch CUDA tensors out, through the
C ABI of libhsefr (include/hsefr.h "Per-kernel entry points").  Used by the unit parity tests
and by the identification stage; the engine calls the same launchers internally."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import _lib
from .lowering import ACT_NONE, ACT_RELU, ACT_RELU6, ACT_SIGMOID, tf_same_padding  # noqa: F401


def _device_guarded(fn):
    """Run the launch with the device of the first CUDA tensor argument current (stream + launch target = the
    device that owns the pointers); mixing devices in one call is an error, not a memory fault."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        devs = [a.device for a in list(args) + list(kw.values()) if hasattr(a, "is_cuda") and a.is_cuda]
        if not devs:
            return fn(*args, **kw)
        if any(d != devs[0] for d in devs):
            raise ValueError("%s: tensors live on different devices: %s" % (fn.__name__, sorted({str(d) for d in devs})))
        with _lib.on_device(devs[0]):
            return fn(*args, **kw)
    return wrapper


def _f32c(t, name):
    torch = _lib.require_gpu()
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous float32 CUDA tensor" % name)
    return t


def _same(h: int, w: int, k: int, stride: int) -> Tuple[int, int, int, int]:
    oh, pt = tf_same_padding(h, k, stride)
    ow, pl = tf_same_padding(w, k, stride)
    return oh, ow, pt, pl


@_device_guarded
def conv3x3_c3(x, w_hwio, shift, stride: int = 2, act: int = ACT_RELU6):
    """Conv2D 3x3 SAME over a 3-channel NHWC image + shift + act (graph nodes #30-34)."""
    torch = _lib.require_gpu()
    _f32c(x, "x"), _f32c(w_hwio, "w"), _f32c(shift, "shift")
    n, h, w, c = x.shape
    if c != 3 or tuple(w_hwio.shape[:3]) != (3, 3, 3):
        raise ValueError("conv3x3_c3 wants x [n,h,w,3] and w [3,3,3,cout]")
    cout = w_hwio.shape[3]
    oh, ow, pt, pl = _same(h, w, 3, stride)
    y = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_conv_c3_bias_act(x.data_ptr(), w_hwio.data_ptr(), shift.data_ptr(), y.data_ptr(),
                                                 n, h, w, 3, 3, stride, pt, pl, oh, ow, cout, act,
                                                 _lib.current_stream_ptr()), "hsefr_conv_c3_bias_act")
    return y


@_device_guarded
def dwconv3x3(x, w_hwc, scale, shift, stride: int = 1, act: int = ACT_RELU6):
    """DepthwiseConv2dNative 3x3 SAME + scale + shift + act (graph nodes #35-39,#44)."""
    torch = _lib.require_gpu()
    _f32c(x, "x"), _f32c(w_hwc, "w"), _f32c(scale, "scale"), _f32c(shift, "shift")
    n, h, w, c = x.shape
    oh, ow, pt, pl = _same(h, w, 3, stride)
    y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_dwconv3x3_bn_relu6(x.data_ptr(), w_hwc.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                   y.data_ptr(), n, h, w, c, stride, pt, pl, oh, ow, act,
                                                   _lib.current_stream_ptr()), "hsefr_dwconv3x3_bn_relu6")
    return y


def split_rows_encode(x, a_log2: int = 12):
    """fp32 [..., c] (c % 32 == 0) -> the split-row image of it as a float16 tensor [..., c/32, 2, 32] ([..., 0, :] = hi,
    [..., 1, :] = lo), computed with torch ops exactly as the kernels do: v = x * 2^a_log2, hi = f16(v), lo = f16(v - hi).
    A reference / container helper for tests; the product path writes this format in dwconv.hip."""
    torch = _lib.require_gpu()
    c = x.shape[-1]
    v = (x.float() * float(2 ** a_log2)).reshape(tuple(x.shape[:-1]) + (c // 32, 32))
    hi = v.to(torch.float16)
    lo = (v - hi.float()).to(torch.float16)
    return torch.stack([hi, lo], dim=-2).contiguous()


def split_rows_decode(xs, a_log2: int = 12):
    """Inverse view of split rows: float16 [..., c/32, 2, 32] -> fp32 [..., c] = (hi + lo) / 2^a_log2 (exact in fp32
    for values the split represents exactly; otherwise within 2^-22 relative)."""
    v = xs[..., 0, :].float() + xs[..., 1, :].float()
    return (v / float(2 ** a_log2)).reshape(tuple(xs.shape[:-3]) + (xs.shape[-3] * 32,))


@_device_guarded
def dwconv3x3_split(x, w_hwc, scale, shift, stride: int = 1, act: int = ACT_RELU6, a_log2: int = 12):
    """dwconv3x3 with its result stored PRE-SPLIT for the GEMM behind it (csrc/dwconv.hip SPLIT): float16 tensor
    [n, oh, ow, c/32, 2, 32] (same bytes as the fp32 tensor)."""
    torch = _lib.require_gpu()
    _f32c(x, "x"), _f32c(w_hwc, "w"), _f32c(scale, "scale"), _f32c(shift, "shift")
    n, h, w, c = x.shape
    oh, ow, pt, pl = _same(h, w, 3, stride)
    y = torch.empty((n, oh, ow, max(c // 32, 0), 2, 32), dtype=torch.float16, device=x.device)
    _lib.check(_lib.lib().hsefr_dwconv3x3_bn_relu6_split(x.data_ptr(), w_hwc.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                         y.data_ptr(), n, h, w, c, stride, pt, pl, oh, ow, act, a_log2,
                                                         _lib.current_stream_ptr()), "hsefr_dwconv3x3_bn_relu6_split")
    return y


@_device_guarded
def pwconv1x1_presplit(xs, w_t, shift, act: int = ACT_RELU6, a_log2: int = 12, prepared=None):
    """1x1 conv + shift + act on PRE-SPLIT activations xs = float16 [..., k/32, 2, 32] (csrc/pwconv_ps.hip); weights as for
    pwconv1x1_f16split (a_log2 must be the exponent xs was split with)."""
    torch = _lib.require_gpu()
    _f32c(shift, "shift")
    if not (xs.is_cuda and xs.dtype == torch.float16 and xs.is_contiguous() and xs.dim() >= 3 and tuple(xs.shape[-2:]) == (2, 32)):
        raise ValueError("xs must be a contiguous float16 CUDA tensor [..., k/32, 2, 32]")
    d_img, d_ds = prepared if prepared is not None else split_weights_device(w_t, xs.device, a_log2)
    k = xs.shape[-3] * 32
    cout = d_img.shape[0]
    m = xs.numel() // (2 * k)
    y = torch.empty(tuple(xs.shape[:-3]) + (cout,), dtype=torch.float32, device=xs.device)
    _lib.check(_lib.lib().hsefr_pwconv1x1_presplit(xs.data_ptr(), d_img.data_ptr(), d_ds.data_ptr(), shift.data_ptr(), y.data_ptr(),
                                                   m, k, cout, act, _lib.current_stream_ptr()), "hsefr_pwconv1x1_presplit")
    return y


@_device_guarded
def pwconv1x1_presplit_gap(xs, w_t, shift, act: int = ACT_RELU6, a_log2: int = 12, prepared=None):
    """pwconv1x1_presplit with the global average pool in its epilogue (csrc/pwconv_ps.hip): xs = split rows [n, h, w, k/32, 2, 32]
    with 33 <= h * w <= 288 -> fp32 [n, cout] means; the pointwise tensor is never written."""
    torch = _lib.require_gpu()
    _f32c(shift, "shift")
    if not (xs.is_cuda and xs.dtype == torch.float16 and xs.is_contiguous() and xs.dim() == 6 and tuple(xs.shape[-2:]) == (2, 32)):
        raise ValueError("xs must be a contiguous float16 CUDA tensor [n, h, w, k/32, 2, 32]")
    d_img, d_ds = prepared if prepared is not None else split_weights_device(w_t, xs.device, a_log2)
    n, h, w = int(xs.shape[0]), int(xs.shape[1]), int(xs.shape[2])
    k, cout = xs.shape[3] * 32, d_img.shape[0]
    y = torch.empty((n, cout), dtype=torch.float32, device=xs.device)
    _lib.check(_lib.lib().hsefr_pwconv1x1_presplit_gap(xs.data_ptr(), d_img.data_ptr(), d_ds.data_ptr(), shift.data_ptr(), y.data_ptr(),
                                                       n * h * w, k, cout, act, h * w, _lib.current_stream_ptr()), "hsefr_pwconv1x1_presplit_gap")
    return y


@_device_guarded
def pwconv1x1_presplit_dw(xs, w_t, shift, dw_w_hwc, dw_scale, dw_shift, act: int = ACT_RELU6, a_log2: int = 12, out_log2: int = 12, prepared=None,
                          dw_stride: int = 1):
    """pwconv1x1_presplit with the NEXT block's depthwise 3x3 / stride 1 / SAME + scale + shift + ReLU6 in its epilogue
    (csrc/pwconv_ps.hip, DW = true).  xs = split rows [n, h, w, k/32, 2, 32] with h * w <= 288; dw_w_hwc [3, 3, cout].
    Returns the depthwise result as split rows [n, h/s, w/s, cout/32, 2, 32] scaled by 2^out_log2 (split_rows_decode); dw_stride 2
    (12x12 maps only) is TF SAME on an even map: no top / left padding."""
    torch = _lib.require_gpu()
    _f32c(shift, "shift"), _f32c(dw_w_hwc, "dw_w"), _f32c(dw_scale, "dw_scale"), _f32c(dw_shift, "dw_shift")
    if not (xs.is_cuda and xs.dtype == torch.float16 and xs.is_contiguous() and xs.dim() == 6 and tuple(xs.shape[-2:]) == (2, 32)):
        raise ValueError("xs must be a contiguous float16 CUDA tensor [n, h, w, k/32, 2, 32]")
    d_img, d_ds = prepared if prepared is not None else split_weights_device(w_t, xs.device, a_log2)
    n, h, w = int(xs.shape[0]), int(xs.shape[1]), int(xs.shape[2])
    k, cout = xs.shape[3] * 32, d_img.shape[0]
    if tuple(dw_w_hwc.shape) != (3, 3, cout) or dw_scale.numel() != cout or dw_shift.numel() != cout:
        raise ValueError("depthwise operands must be [3,3,%d], [%d], [%d]" % (cout, cout, cout))
    amp = float(2 ** out_log2)
    consts = torch.cat([dw_w_hwc.reshape(9, cout), (dw_scale * amp).reshape(1, cout), (dw_shift * amp).reshape(1, cout)]).contiguous()
    ys = torch.empty((n, h // dw_stride, w // dw_stride, cout // 32, 2, 32), dtype=torch.float16, device=xs.device)
    _lib.check(_lib.lib().hsefr_pwconv1x1_presplit_dw(xs.data_ptr(), d_img.data_ptr(), d_ds.data_ptr(), shift.data_ptr(), consts.data_ptr(),
                                                      ys.data_ptr(), n * h * w, k, cout, act, w, h * w, dw_stride, out_log2,
                                                      _lib.current_stream_ptr()),
               "hsefr_pwconv1x1_presplit_dw")
    return ys


@_device_guarded
def pwconv1x1(x, w_t, shift, act: int = ACT_RELU6):
    """1x1 conv + shift + act on NHWC x [..., k]; w_t is the TF kernel transposed: [cout, k]."""
    torch = _lib.require_gpu()
    _f32c(x, "x"), _f32c(w_t, "w_t"), _f32c(shift, "shift")
    k = x.shape[-1]
    cout = w_t.shape[0]
    m = x.numel() // k
    y = torch.empty(tuple(x.shape[:-1]) + (cout,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_pwconv1x1_bias_relu6(x.data_ptr(), w_t.data_ptr(), shift.data_ptr(), y.data_ptr(),
                                                     m, k, cout, act, _lib.current_stream_ptr()),
               "hsefr_pwconv1x1_bias_relu6")
    return y


def split_weights_device(w_t, device, a_log2: int = 12):
    """Host-side split of a pointwise kernel [cout, k] into the device image + descale the f16-split GEMM takes."""
    torch = _lib.require_gpu()
    from . import lowering
    w_np = w_t.detach().cpu().numpy() if hasattr(w_t, "detach") else w_t
    img, descale = lowering.split_pointwise_weights(w_np, a_log2)
    return torch.from_numpy(img.view(np.int16)).to(device), torch.from_numpy(descale).to(device)


@_device_guarded
def pwconv1x1_f16split(x, w_t, shift, act: int = ACT_RELU6, a_log2: int = 12, prepared=None):
    """1x1 conv + shift + act with split-f16 products (fp32-grade, csrc/pwconv_f16s.hip).  PRECONDITION:
    |x| * 2^a_log2 < 32768 (a ReLU6 producer with the default 12).  w_t [cout, k] fp32 is split on the host
    (or pass prepared=split_weights_device(...))."""
    torch = _lib.require_gpu()
    _f32c(x, "x"), _f32c(shift, "shift")
    d_img, d_ds = prepared if prepared is not None else split_weights_device(w_t, x.device, a_log2)
    k = x.shape[-1]
    cout = d_img.shape[0]
    m = x.numel() // k
    y = torch.empty(tuple(x.shape[:-1]) + (cout,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_pwconv1x1_f16split(x.data_ptr(), d_img.data_ptr(), d_ds.data_ptr(), shift.data_ptr(),
                                                   y.data_ptr(), m, k, cout, a_log2, act, _lib.current_stream_ptr()),
               "hsefr_pwconv1x1_f16split")
    return y


@_device_guarded
def dwpw_f16split(x, w_hwc, dscale, dshift, wp_t, pshift, stride: int = 1, act: int = ACT_RELU6, a_log2: int = 12, prepared=None):
    """One MobileNet block in one kernel, any c % 32 == 0 / cout % 64 == 0: depthwise 3x3 SAME + scale + shift + ReLU6 ->
    pointwise 1x1 + shift + act with split-f16 products (csrc/dwpw_f16s.hip).  wp_t [cout, c] fp32 is split on the host."""
    torch = _lib.require_gpu()
    for t, nm in ((x, "x"), (w_hwc, "w"), (dscale, "dscale"), (dshift, "dshift"), (pshift, "pshift")):
        _f32c(t, nm)
    d_img, d_ds = prepared if prepared is not None else split_weights_device(wp_t, x.device, a_log2)
    n, h, w, c = x.shape
    cout = d_img.shape[0]
    oh, ow, pt, pl = _same(h, w, 3, stride)
    y = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_dwpw_f16split(x.data_ptr(), w_hwc.data_ptr(), dscale.data_ptr(), dshift.data_ptr(), d_img.data_ptr(),
                                              d_ds.data_ptr(), pshift.data_ptr(), y.data_ptr(), n, h, w, c, stride, pt, pl, oh, ow,
                                              cout, a_log2, act, _lib.current_stream_ptr()), "hsefr_dwpw_f16split")
    return y


@_device_guarded
def stem_fused(x, conv_w, conv_shift, w_hwc, dscale, dshift, wp_t, pshift, act: int = ACT_RELU6, a_log2: int = 12, prepared=None):
    """The MobileNet stem in one kernel (csrc/stem_fused.hip): conv 3x3/2 SAME 3->32 + shift + ReLU6 -> depthwise 3x3/1 +
    scale + shift + ReLU6 -> pointwise 32->64 + shift + act.  conv_w TF HWIO [3,3,3,32], w_hwc [3,3,32], wp_t [64,32]."""
    torch = _lib.require_gpu()
    if not hasattr(_lib.lib(), "hsefr_stem_fused"):
        raise NotImplementedError("round 1's fused stem is part of DEVELOPMENT builds of the library only (HSEFR_LIB=libhsefr_dev.so)")
    for t, nm in ((x, "x"), (conv_w, "conv_w"), (conv_shift, "conv_shift"), (w_hwc, "w"), (dscale, "dscale"), (dshift, "dshift"),
                  (pshift, "pshift")):
        _f32c(t, nm)
    d_img, d_ds = prepared if prepared is not None else split_weights_device(wp_t, x.device, a_log2)
    n, h, w, c = x.shape
    if c != 3 or tuple(conv_w.shape) != (3, 3, 3, 32) or tuple(w_hwc.shape) != (3, 3, 32) or d_img.shape[0] != 64:
        raise NotImplementedError("stem_fused covers 3 -> 32 -> 64 channels")
    oh, ow, pt, pl = _same(h, w, 3, 2)
    y = torch.empty((n, oh, ow, 64), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_stem_fused(x.data_ptr(), conv_w.data_ptr(), conv_shift.data_ptr(), w_hwc.data_ptr(), dscale.data_ptr(),
                                           dshift.data_ptr(), d_img.data_ptr(), d_ds.data_ptr(), pshift.data_ptr(), y.data_ptr(),
                                           n, h, w, pt, pl, oh, ow, a_log2, act, _lib.current_stream_ptr()), "hsefr_stem_fused")
    return y


@_device_guarded
def stem2_fused(x, conv_w, conv_shift, w1_hwc, d1scale, d1shift, wp_t, pshift, w2_hwc, d2scale, d2shift, act: int = ACT_RELU6,
                a_log2: int = 12, prepared=None):
    """Stem + the depthwise of block 2 in one kernel (csrc/stem2_fused.hip): conv 3x3/2 3->32 -> depthwise 3x3/1 -> pointwise
    32->64 (all + ReLU6) -> depthwise 3x3/2 + scale + shift + act.  w2_hwc [3,3,64]; output [n, ceil(h/4), ceil(w/4), 64]."""
    torch = _lib.require_gpu()
    for t, nm in ((x, "x"), (conv_w, "conv_w"), (conv_shift, "conv_shift"), (w1_hwc, "w1"), (d1scale, "d1scale"), (d1shift, "d1shift"),
                  (pshift, "pshift"), (w2_hwc, "w2"), (d2scale, "d2scale"), (d2shift, "d2shift")):
        _f32c(t, nm)
    d_img, d_ds = prepared if prepared is not None else split_weights_device(wp_t, x.device, a_log2)
    n, h, w, c = x.shape
    if c != 3 or tuple(conv_w.shape) != (3, 3, 3, 32) or tuple(w1_hwc.shape) != (3, 3, 32) or d_img.shape[0] != 64 or \
            tuple(w2_hwc.shape) != (3, 3, 64):
        raise NotImplementedError("stem2_fused covers 3 -> 32 -> 64 channels")
    h1, w1, pt, pl = _same(h, w, 3, 2)
    oh2, ow2, pt2, pl2 = _same(h1, w1, 3, 2)
    y = torch.empty((n, oh2, ow2, 64), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_stem2_fused(x.data_ptr(), conv_w.data_ptr(), conv_shift.data_ptr(), w1_hwc.data_ptr(), d1scale.data_ptr(),
                                            d1shift.data_ptr(), d_img.data_ptr(), d_ds.data_ptr(), pshift.data_ptr(), w2_hwc.data_ptr(),
                                            d2scale.data_ptr(), d2shift.data_ptr(), y.data_ptr(), n, h, w, pt, pl, h1, w1, pt2, pl2,
                                            oh2, ow2, a_log2, act, _lib.current_stream_ptr()), "hsefr_stem2_fused")
    return y


@_device_guarded
def stem3_fused(x, conv_w, conv_shift, w1_hwc, d1scale, d1shift, wp_t, pshift, w2_hwc, d2scale, d2shift, act: int = ACT_RELU6,
                a_log2: int = 12, in_log2: int = 7, prepared=None, overflow=None):
    """stem2_fused for an input with the declared bound |x| < 2^(15 - in_log2) (csrc/stem3_fused.hip): conv1 on the f16 MFMA
    from two-term splits.  overflow: optional int32 CUDA tensor [1] that is OR-ed with 1 when a value breaks the bound."""
    torch = _lib.require_gpu()
    from . import lowering
    for t, nm in ((x, "x"), (conv_shift, "conv_shift"), (w1_hwc, "w1"), (d1scale, "d1scale"), (d1shift, "d1shift"),
                  (pshift, "pshift"), (w2_hwc, "w2"), (d2scale, "d2scale"), (d2shift, "d2shift")):
        _f32c(t, nm)
    d_img, d_ds = prepared if prepared is not None else split_weights_device(wp_t, x.device, a_log2)
    n, h, w, c = x.shape
    cw = conv_w.detach().cpu().numpy() if hasattr(conv_w, "detach") else np.asarray(conv_w)
    if c != 3 or tuple(cw.shape) != (3, 3, 3, 32) or tuple(w1_hwc.shape) != (3, 3, 32) or d_img.shape[0] != 64 or tuple(w2_hwc.shape) != (3, 3, 64):
        raise NotImplementedError("stem3_fused covers 3 -> 32 -> 64 channels")
    cw_t = np.zeros((32, 32), np.float32)
    cw_t[:, :27] = cw.reshape(27, 32).T
    cimg, cds = lowering.split_pointwise_weights(cw_t, in_log2)
    d_cimg = torch.from_numpy(cimg.view(np.int16)).to(x.device)
    d_cds = torch.from_numpy(cds).to(x.device)
    h1, w1, pt, pl = _same(h, w, 3, 2)
    oh2, ow2, pt2, pl2 = _same(h1, w1, 3, 2)
    y = torch.empty((n, oh2, ow2, 64), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_stem3_fused(x.data_ptr(), d_cimg.data_ptr(), d_cds.data_ptr(), conv_shift.data_ptr(), w1_hwc.data_ptr(),
                                            d1scale.data_ptr(), d1shift.data_ptr(), d_img.data_ptr(), d_ds.data_ptr(), pshift.data_ptr(),
                                            w2_hwc.data_ptr(), d2scale.data_ptr(), d2shift.data_ptr(), y.data_ptr(),
                                            None if overflow is None else overflow.data_ptr(), n, h, w, pt, pl, h1, w1, pt2, pl2,
                                            oh2, ow2, in_log2, a_log2, act, _lib.current_stream_ptr()), "hsefr_stem3_fused")
    return y


def stem5_stream(*args, **kw):
    """stem4_fused's four layers, operands and bits as the streaming kernel of round 4 (csrc/stem5_stream.hip): same arguments."""
    return stem4_fused(*args, _entry="hsefr_stem5_stream", **kw)


@_device_guarded
def stem4_fused(x, conv_w, conv_shift, w1_hwc, d1scale, d1shift, wp_t, pshift, w2_hwc, d2scale, d2shift, act: int = ACT_RELU6,
                a_log2: int = 12, in_log2: int = 7, prepared=None, overflow=None, u8_mean_bgr=None, _entry: str = "hsefr_stem4_fused"):
    """stem3_fused for inputs whose edges are multiples of 4 (csrc/stem4_fused.hip): no im2col, conv1's operands straight from
    the f16 window.  x float32 [n,h,w,3] within the declared bound -- or, with u8_mean_bgr, the RESIZED image as uint8 RGB
    [n,h,w,3]: float conversion, channel reversal and the BGR mean are folded into the constants (in_log2 is then 0)."""
    torch = _lib.require_gpu()
    from . import lowering
    for t, nm in ((conv_shift, "conv_shift"), (w1_hwc, "w1"), (d1scale, "d1scale"), (d1shift, "d1shift"),
                  (pshift, "pshift"), (w2_hwc, "w2"), (d2scale, "d2scale"), (d2shift, "d2shift")):
        _f32c(t, nm)
    u8 = u8_mean_bgr is not None
    if u8:
        if x.dtype != torch.uint8 or not x.is_cuda or not x.is_contiguous():
            raise ValueError("x: expected a contiguous uint8 CUDA tensor")
    else:
        _f32c(x, "x")
    d_img, d_ds = prepared if prepared is not None else split_weights_device(wp_t, x.device, a_log2)
    n, h, w, c = x.shape
    cw = conv_w.detach().cpu().numpy() if hasattr(conv_w, "detach") else np.asarray(conv_w)
    if c != 3 or tuple(cw.shape) != (3, 3, 3, 32) or tuple(w1_hwc.shape) != (3, 3, 32) or d_img.shape[0] != 64 or tuple(w2_hwc.shape) != (3, 3, 64):
        raise NotImplementedError("stem4_fused covers 3 -> 32 -> 64 channels")
    if u8:
        img4, ds4 = lowering.stem4_conv_image(cw, 0, reverse_channels=True)
        sh = lowering.stem4_u8_shifts(cw, conv_shift.detach().cpu().numpy(), u8_mean_bgr)
        d_sh = torch.from_numpy(sh).to(x.device)
        in_log2 = 0
    else:
        img4, ds4 = lowering.stem4_conv_image(cw, in_log2)
        d_sh = conv_shift
    d_cimg = torch.from_numpy(img4.view(np.int16)).to(x.device)
    d_cds = torch.from_numpy(ds4).to(x.device)
    y = torch.empty((n, h // 4, w // 4, 64), dtype=torch.float32, device=x.device)
    _lib.check(getattr(_lib.lib(), _entry)(x.data_ptr(), 1 if u8 else 0, d_cimg.data_ptr(), d_cds.data_ptr(), d_sh.data_ptr(), w1_hwc.data_ptr(),
                                            d1scale.data_ptr(), d1shift.data_ptr(), d_img.data_ptr(), d_ds.data_ptr(), pshift.data_ptr(),
                                            w2_hwc.data_ptr(), d2scale.data_ptr(), d2shift.data_ptr(), y.data_ptr(),
                                            None if overflow is None else overflow.data_ptr(), n, h, w, in_log2, a_log2, act,
                                            _lib.current_stream_ptr()), _entry)
    return y


@_device_guarded
def dwpw_fused(x, w_hwc, dscale, dshift, wp_t, pshift, stride: int = 1):
    """One early MobileNet block in one kernel: depthwise 3x3 SAME + scale + shift + ReLU6 -> pointwise 1x1 + shift +
    ReLU6 (graph nodes #35-#49).  c in {32, 64}, cout in {64, 128}; wp_t is the pointwise kernel transposed [cout, c]."""
    torch = _lib.require_gpu()
    for t, nm in ((x, "x"), (w_hwc, "w"), (dscale, "dscale"), (dshift, "dshift"), (wp_t, "wp_t"), (pshift, "pshift")):
        _f32c(t, nm)
    n, h, w, c = x.shape
    cout = wp_t.shape[0]
    oh, ow, pt, pl = _same(h, w, 3, stride)
    y = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_dwpw_fused(x.data_ptr(), w_hwc.data_ptr(), dscale.data_ptr(), dshift.data_ptr(), wp_t.data_ptr(),
                                           pshift.data_ptr(), y.data_ptr(), n, h, w, c, stride, pt, pl, oh, ow, cout,
                                           _lib.current_stream_ptr()), "hsefr_dwpw_fused")
    return y


@_device_guarded
def gap(x):
    torch = _lib.require_gpu()
    _f32c(x, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_gap(x.data_ptr(), y.data_ptr(), n, h * w, c, _lib.current_stream_ptr()), "hsefr_gap")
    return y


@_device_guarded
def dense(x, w, bias=None, act: int = ACT_NONE):
    torch = _lib.require_gpu()
    _f32c(x, "x"), _f32c(w, "w")
    n, k = x.shape
    cout = w.shape[1]
    y = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_dense(x.data_ptr(), w.data_ptr(), None if bias is None else _f32c(bias, "bias").data_ptr(),
                                      y.data_ptr(), n, k, cout, act, _lib.current_stream_ptr()), "hsefr_dense")
    return y


@_device_guarded
def softmax(x):
    torch = _lib.require_gpu()
    _f32c(x, "x")
    n, c = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().hsefr_softmax(x.data_ptr(), y.data_ptr(), n, c, _lib.current_stream_ptr()), "hsefr_softmax")
    return y


@_device_guarded
def heads_fused(x, w1, b1, wa, ba, wg, bg):
    """The age / gender heads in one launch (hsefr_heads_fused; facial_analysis.py:109): x [n,k] -> (hidden [n,256] = relu(x.w1 + b1),
    logits [n,a] = hidden.wa + ba, age_probs = softmax(logits), gender [n,1] = sigmoid(hidden.wg + bg)); w1 [k,256], wa [256,a], wg [256,1]."""
    torch = _lib.require_gpu()
    for t, nm in ((x, "x"), (w1, "w1"), (b1, "b1"), (wa, "wa"), (ba, "ba"), (wg, "wg"), (bg, "bg")):
        _f32c(t, nm)
    n, k = x.shape
    a = wa.shape[1]
    if tuple(w1.shape) != (k, 256) or wa.shape[0] != 256 or tuple(wg.shape) != (256, 1):
        raise ValueError("heads_fused: w1 [k,256], wa [256,a], wg [256,1]")
    hidden = torch.empty((n, 256), dtype=torch.float32, device=x.device)
    logits = torch.empty((n, a), dtype=torch.float32, device=x.device)
    probs = torch.empty((n, a), dtype=torch.float32, device=x.device)
    gender = torch.empty((n, 1), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_heads_fused(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), wa.data_ptr(), ba.data_ptr(), wg.data_ptr(), bg.data_ptr(),
                                            hidden.data_ptr(), logits.data_ptr(), probs.data_ptr(), gender.data_ptr(), n, k, a,
                                            _lib.current_stream_ptr()), "hsefr_heads_fused")
    return hidden, logits, probs, gender


@_device_guarded
def l2_normalize(x):
    """preprocessing.normalize(X, norm='l2') (facerec_test.py:401)."""
    torch = _lib.require_gpu()
    _f32c(x, "x")
    n, d = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().hsefr_l2_normalize(x.data_ptr(), y.data_ptr(), n, d, _lib.current_stream_ptr()),
               "hsefr_l2_normalize")
    return y


@_device_guarded
def nn1(queries, gallery):
    """Index (int32) and squared L2 distance of each query's nearest gallery row."""
    torch = _lib.require_gpu()
    _f32c(queries, "queries"), _f32c(gallery, "gallery")
    nq, d = queries.shape
    ng = gallery.shape[0]
    if gallery.shape[1] != d:
        raise ValueError("queries are %d-D, gallery is %d-D" % (d, gallery.shape[1]))
    idx = torch.empty((nq,), dtype=torch.int32, device=queries.device)
    dist = torch.empty((nq,), dtype=torch.float32, device=queries.device)
    _lib.check(_lib.lib().hsefr_nn1(queries.data_ptr(), gallery.data_ptr(), nq, ng, d, idx.data_ptr(), dist.data_ptr(),
                                    _lib.current_stream_ptr()), "hsefr_nn1")
    return idx, dist


@_device_guarded
def conv2d_direct(x, w_hwio, bias=None, alpha=None, stride: int = 1, padding: str = "VALID"):
    """Generic Conv2D + BiasAdd + optional PReLU (MTCNN nets).  padding: 'VALID' | 'SAME' (TensorFlow rule)."""
    torch = _lib.require_gpu()
    _f32c(x, "x"), _f32c(w_hwio, "w")
    n, h, w, c = x.shape
    kh, kw, wc, cout = w_hwio.shape
    if wc != c:
        raise ValueError("kernel expects %d input channels, tensor has %d" % (wc, c))
    if padding == "SAME":
        oh, pt = tf_same_padding(h, kh, stride)
        ow, pl = tf_same_padding(w, kw, stride)
    else:
        oh, ow, pt, pl = (h - kh) // stride + 1, (w - kw) // stride + 1, 0, 0
    y = torch.empty((n, max(oh, 0), max(ow, 0), cout), dtype=torch.float32, device=x.device)
    if y.numel():
        _lib.check(_lib.lib().hsefr_conv2d_direct(x.data_ptr(), w_hwio.data_ptr(), None if bias is None else bias.data_ptr(),
                                                  None if alpha is None else alpha.data_ptr(), y.data_ptr(), n, h, w, c, oh, ow,
                                                  cout, kh, kw, stride, pt, pl, _lib.current_stream_ptr()), "hsefr_conv2d_direct")
    return y


@_device_guarded
def conv2d_f32(x, w_hwio, scale=None, shift=None, stride: int = 1, pad: int = 0, res=None, act: int = ACT_NONE, mfma: bool = False):
    """General Conv2D in exact fp32 + per-channel scale / shift + optional residual + activation (the fp32-grade mode of the
    ResNet-style graphs).  x [n,h,w,c], w_hwio [kh,kw,c,cout] with cout % 4 == 0, symmetric zero padding `pad`.
    mfma: the implicit GEMM on the fp32 matrix pipe (csrc/conv_f32_mfma.hip, cout % 64 == 0; what the engine runs when it covers
    the layer) instead of the vector-FMA direct convolution."""
    torch = _lib.require_gpu()
    _f32c(x, "x"), _f32c(w_hwio, "w")
    n, h, w, c = x.shape
    kh, kw, wc, cout = w_hwio.shape
    if wc != c:
        raise ValueError("kernel expects %d input channels, tensor has %d" % (wc, c))
    oh, ow = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    if oh <= 0 or ow <= 0:
        raise ValueError("kernel %dx%d does not fit a %dx%d tensor padded by %d" % (kh, kw, h, w, pad))
    y = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
    if res is not None and tuple(_f32c(res, "res").shape) != tuple(y.shape):
        raise ValueError("residual shape %r, output shape %r" % (tuple(res.shape), tuple(y.shape)))
    for v, name in ((scale, "scale"), (shift, "shift")):
        if v is not None and _f32c(v, name).numel() != cout:
            raise ValueError("%s has %d elements for %d output channels" % (name, v.numel(), cout))
    fn = _lib.lib().hsefr_conv2d_f32_mfma if mfma else _lib.lib().hsefr_conv2d_f32
    _lib.check(fn(x.data_ptr(), w_hwio.data_ptr(), None if scale is None else scale.data_ptr(),
                                           None if shift is None else shift.data_ptr(), None if res is None else res.data_ptr(),
                                           y.data_ptr(), n, h, w, c, oh, ow, cout, kh, kw, stride, pad, pad, act,
                                           _lib.current_stream_ptr()), "hsefr_conv2d_f32")
    return y


@_device_guarded
def maxpool(x, k: int, stride: int, padding: str = "SAME"):
    torch = _lib.require_gpu()
    _f32c(x, "x")
    n, h, w, c = x.shape
    if padding == "SAME":
        oh, pt = tf_same_padding(h, k, stride)
        ow, pl = tf_same_padding(w, k, stride)
    else:
        oh, ow, pt, pl = (h - k) // stride + 1, (w - k) // stride + 1, 0, 0
    y = torch.empty((n, max(oh, 0), max(ow, 0), c), dtype=torch.float32, device=x.device)
    if y.numel():
        _lib.check(_lib.lib().hsefr_maxpool_f32(x.data_ptr(), y.data_ptr(), n, h, w, c, oh, ow, k, stride, pt, pl,
                                                _lib.current_stream_ptr()), "hsefr_maxpool_f32")
    return y


@_device_guarded
def pairwise_distances(x, y=None):
    """sklearn pairwise_distances(X[, Y]) (euclidean) -> CUDA float32 [n, m] (facial_clustering_test.py:396)."""
    torch = _lib.require_gpu()
    _f32c(x, "x")
    y = x if y is None else _f32c(y, "y")
    n, d = x.shape
    m = y.shape[0]
    out = torch.empty((n, m), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_pairwise_dist(x.data_ptr(), y.data_ptr(), n, m, d, out.data_ptr(), _lib.current_stream_ptr()),
               "hsefr_pairwise_dist")
    return out


# ---- bf16 ResNet-50 kernels -------------------------------------------------------------------------
def _bf16c(t, name):
    torch = _lib.require_gpu()
    if not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous bfloat16 CUDA tensor" % name)
    return t


def bf16_from_bits(bits_u16, device=None):
    """NumPy uint16 bf16 bit patterns -> CUDA bfloat16 tensor (a container; no arithmetic)."""
    torch = _lib.require_gpu()
    import numpy as np
    return torch.from_numpy(np.ascontiguousarray(bits_u16).view(np.int16)).to(_lib.cuda_device(device)).view(torch.bfloat16)


@_device_guarded
def conv_bf16(x, w_packed, scale, shift, kh: int, kw: int, stride: int = 1, pad: int = 0, res=None, act: int = ACT_RELU):
    """x [n,h,w,c] bf16; w_packed [cout, kh*kw*c] bf16 (resnet50.pack_conv_weight); -> [n,oh,ow,cout] bf16."""
    torch = _lib.require_gpu()
    _bf16c(x, "x"), _bf16c(w_packed, "w"), _f32c(scale, "scale"), _f32c(shift, "shift")
    n, h, w, c = x.shape
    cout = w_packed.shape[0]
    oh, ow = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    y = torch.empty((n, oh, ow, cout), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().hsefr_conv_bf16(x.data_ptr(), w_packed.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                          None if res is None else _bf16c(res, "res").data_ptr(), y.data_ptr(),
                                          n, h, w, c, oh, ow, cout, kh, kw, stride, pad, pad, act,
                                          _lib.current_stream_ptr()), "hsefr_conv_bf16")
    return y


@_device_guarded
def conv1x1_proj_bf16(x, w_packed, scale, shift, x2, w2_packed, scale2, shift2, stride2: int = 1, act: int = ACT_RELU):
    """The increase layer of a ResNet stage's first block with its projected shortcut in one launch (hsefr_conv1x1_proj_bf16):
    act(bf16(scale * x.w + shift) + bf16(scale2 * x2[::stride2, ::stride2].w2 + shift2)).  x [n,oh,ow,c] bf16, w_packed [cout,c],
    x2 [n,h2,w2,c2] bf16 (the block input), w2_packed [cout,c2] -> [n,oh,ow,cout] bf16."""
    torch = _lib.require_gpu()
    _bf16c(x, "x"), _bf16c(w_packed, "w"), _f32c(scale, "scale"), _f32c(shift, "shift")
    _bf16c(x2, "x2"), _bf16c(w2_packed, "w2"), _f32c(scale2, "scale2"), _f32c(shift2, "shift2")
    n, oh, ow, c = x.shape
    n2, h2, w2, c2 = x2.shape
    cout = w_packed.shape[0]
    if n2 != n or tuple(w2_packed.shape) != (cout, c2) or w_packed.shape[1] != c:
        raise ValueError("conv1x1_proj_bf16: inconsistent shapes")
    y = torch.empty((n, oh, ow, cout), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().hsefr_conv1x1_proj_bf16(x.data_ptr(), w_packed.data_ptr(), scale.data_ptr(), shift.data_ptr(), x2.data_ptr(),
                                                  w2_packed.data_ptr(), scale2.data_ptr(), shift2.data_ptr(), y.data_ptr(), n, oh, ow, c, cout,
                                                  c2, stride2, h2, w2, act, _lib.current_stream_ptr()), "hsefr_conv1x1_proj_bf16")
    return y


@_device_guarded
def conv1x1_sres_bf16(x, w_packed, scale, shift, res, res_stride: int = 2, act: int = ACT_RELU):
    """A 1x1 convolution whose residual is a stride view of a larger map (hsefr_conv1x1_sres_bf16):
    act(bf16(scale * x.w + shift) + res[:, ::res_stride, ::res_stride, :][:, :oh, :ow]).  x [n,oh,ow,c], res [n,h2,w2,cout] bf16."""
    torch = _lib.require_gpu()
    _bf16c(x, "x"), _bf16c(w_packed, "w"), _f32c(scale, "scale"), _f32c(shift, "shift"), _bf16c(res, "res")
    n, oh, ow, c = x.shape
    n2, h2, w2, cout = res.shape
    if n2 != n or tuple(w_packed.shape) != (cout, c):
        raise ValueError("conv1x1_sres_bf16: inconsistent shapes")
    y = torch.empty((n, oh, ow, cout), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().hsefr_conv1x1_sres_bf16(x.data_ptr(), w_packed.data_ptr(), scale.data_ptr(), shift.data_ptr(), res.data_ptr(), y.data_ptr(),
                                                  n, oh, ow, c, cout, res_stride, h2, w2, act, _lib.current_stream_ptr()), "hsefr_conv1x1_sres_bf16")
    return y


def conv1x1_pair_bf16(x, w1_packed, scale1, shift1, w2_packed, scale2, shift2, res=None, x2=None, wp_packed=None, scale_p=None, shift_p=None,
                      act1: int = ACT_RELU, act2: int = ACT_RELU, y1_sub2: bool = False):
    """A bottleneck's increase layer and the next bottleneck's reduce layer in one launch (hsefr_conv1x1_pair_bf16):
    y1 = act1(bf16(scale1 * x.w1 + shift1) + R), R = res or bf16(scale_p * x2.wp + shift_p);  y2 = act2(bf16(scale2 * y1.w2 + shift2)).
    x [n,h,w,c] bf16, w1_packed [cout1,c], res [n,h,w,cout1] | x2 [n,h,w,c2] with wp_packed [cout1,c2], w2_packed [cout2,cout1]
    -> (y1 [n,h,w,cout1], y2 [n,h,w,cout2]) bf16.  y1_sub2: y1 is stored at even rows / columns only, [n,(h+1)//2,(w+1)//2,cout1]
    (hsefr_conv1x1_pair_sub2_bf16)."""
    torch = _lib.require_gpu()
    _bf16c(x, "x"), _bf16c(w1_packed, "w1"), _f32c(scale1, "scale1"), _f32c(shift1, "shift1")
    _bf16c(w2_packed, "w2"), _f32c(scale2, "scale2"), _f32c(shift2, "shift2")
    n, h, w, c = x.shape
    cout1, cout2 = w1_packed.shape[0], w2_packed.shape[0]
    if w1_packed.shape[1] != c or w2_packed.shape[1] != cout1:
        raise ValueError("conv1x1_pair_bf16: inconsistent shapes")
    c2 = 0
    if res is not None:
        _bf16c(res, "res")
        if tuple(res.shape) != (n, h, w, cout1) or x2 is not None:
            raise ValueError("conv1x1_pair_bf16: res must be [n,h,w,cout1], and exclusive with the projected shortcut")
    else:
        _bf16c(x2, "x2"), _bf16c(wp_packed, "wp"), _f32c(scale_p, "scale_p"), _f32c(shift_p, "shift_p")
        c2 = x2.shape[3]
        if tuple(x2.shape[:3]) != (n, h, w) or tuple(wp_packed.shape) != (cout1, c2):
            raise ValueError("conv1x1_pair_bf16: the projected shortcut reads the same pixels")
    y1 = torch.empty((n, (h + 1) // 2, (w + 1) // 2, cout1) if y1_sub2 else (n, h, w, cout1), dtype=torch.bfloat16, device=x.device)
    y2 = torch.empty((n, h, w, cout2), dtype=torch.bfloat16, device=x.device)
    ptr = lambda t: 0 if t is None else t.data_ptr()
    if y1_sub2:
        _lib.check(_lib.lib().hsefr_conv1x1_pair_sub2_bf16(x.data_ptr(), w1_packed.data_ptr(), scale1.data_ptr(), shift1.data_ptr(), ptr(res), ptr(x2),
                                                           ptr(wp_packed), ptr(scale_p), ptr(shift_p), y1.data_ptr(), w2_packed.data_ptr(),
                                                           scale2.data_ptr(), shift2.data_ptr(), y2.data_ptr(), n, h, w, c, cout1, cout2, c2, act1, act2,
                                                           _lib.current_stream_ptr()), "hsefr_conv1x1_pair_sub2_bf16")
        return y1, y2
    _lib.check(_lib.lib().hsefr_conv1x1_pair_bf16(x.data_ptr(), w1_packed.data_ptr(), scale1.data_ptr(), shift1.data_ptr(), ptr(res), ptr(x2),
                                                  ptr(wp_packed), ptr(scale_p), ptr(shift_p), y1.data_ptr(), w2_packed.data_ptr(),
                                                  scale2.data_ptr(), shift2.data_ptr(), y2.data_ptr(), n * h * w, c, cout1, cout2, c2, act1, act2,
                                                  _lib.current_stream_ptr()), "hsefr_conv1x1_pair_bf16")
    return y1, y2


@_device_guarded
def stem7x7_bf16(x, w_packed, scale, shift, act: int = ACT_RELU):
    torch = _lib.require_gpu()
    _f32c(x, "x"), _bf16c(w_packed, "w")
    n, h, w, c = x.shape
    oh, ow = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    y = torch.empty((n, oh, ow, 64), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().hsefr_stem7x7_bf16(x.data_ptr(), w_packed.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                             y.data_ptr(), n, h, w, oh, ow, act, _lib.current_stream_ptr()),
               "hsefr_stem7x7_bf16")
    return y


@_device_guarded
def stem7x7_pool_bf16(x, w_packed, scale, shift, ceil_mode: bool = True, pool_pad: int = 0):
    """conv1 7x7/2 pad 3 (3 -> 64) + scale + shift + ReLU + max-pool 3x3/2 in one kernel: x [n,h,w,3] fp32 -> [n,ph,pw,64] bf16.
    ceil_mode / pool_pad as in maxpool3x3s2_bf16 (Caffe ceil mode pads 0; pool_pad 1 = an explicit Pad(1) + VALID pool)."""
    torch = _lib.require_gpu()
    _f32c(x, "x"), _bf16c(w_packed, "w"), _f32c(scale, "scale"), _f32c(shift, "shift")
    n, h, w, c = x.shape
    if c != 3 or tuple(w_packed.shape) != (64, 256):
        raise ValueError("stem7x7_pool: x must be [n,h,w,3] and the weight image [64,256]")
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if ceil_mode:
        ph, pw = -(-(oh + 2 * pool_pad - 3) // 2) + 1, -(-(ow + 2 * pool_pad - 3) // 2) + 1
    else:
        ph, pw = (oh + 2 * pool_pad - 3) // 2 + 1, (ow + 2 * pool_pad - 3) // 2 + 1
    y = torch.empty((n, ph, pw, 64), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().hsefr_stem7x7_pool_bf16(x.data_ptr(), w_packed.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(),
                                                  n, h, w, ph, pw, pool_pad, pool_pad, _lib.current_stream_ptr()), "hsefr_stem7x7_pool_bf16")
    return y


@_device_guarded
def maxpool3x3s2_bf16(x, ceil_mode: bool = True):
    torch = _lib.require_gpu()
    _bf16c(x, "x")
    n, h, w, c = x.shape
    if ceil_mode:
        oh, ow = -(-(h - 3) // 2) + 1, -(-(w - 3) // 2) + 1
    else:
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    y = torch.empty((n, oh, ow, c), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().hsefr_maxpool3x3s2_bf16(x.data_ptr(), y.data_ptr(), n, h, w, c, oh, ow, 0, 0,
                                                  _lib.current_stream_ptr()), "hsefr_maxpool3x3s2_bf16")
    return y


@_device_guarded
def gap_bf16(x):
    torch = _lib.require_gpu()
    _bf16c(x, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hsefr_gap_bf16(x.data_ptr(), y.data_ptr(), n, h * w, c, _lib.current_stream_ptr()),
               "hsefr_gap_bf16")
    return y
