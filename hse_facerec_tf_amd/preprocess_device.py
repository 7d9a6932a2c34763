"""Device-side image preparation: the resize + BGR + mean steps of
facerec_test.py:80-112 and facial_analysis.py:95-107 on the GPU, bit-exact with PIL / OpenCV's 8-bit paths.

Host work here is table building only (per-axis taps and integer weights, cached per size pair); the
pixels are resampled by hsefr_preprocess_pil_u8 / hsefr_preprocess_cv_u8.
"""
from __future__ import annotations

import ctypes
import math
from functools import lru_cache
from typing import Sequence, Tuple

import numpy as np

from . import _lib
from .preprocess import IMAGENET_CAFFE_BGR_MEAN, VGGFACE2_BGR_MEAN

COLOR_BGR_MEAN_F64, COLOR_RGB_UNIT, COLOR_BGR_MEAN_F32, COLOR_NONE_U8 = 0, 1, 2, 3
_PRECISION_BITS = 32 - 8 - 2      # Pillow: Resample.c


def pil_bilinear_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray, int]:
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1.0):
    -> (first tap [out], tap count [out], int32 weights [out, ksize], ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        ww = 0.0
        for x in range(n):
            a = (x + lo - center + 0.5) * ss
            w = 1.0 - abs(a) if abs(a) < 1.0 else 0.0
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :n] /= ww
        xmin[xx], cnt[xx] = lo, n
    # normalize_coeffs_8bpc: round half away from zero into 22-bit fixed point
    ik = np.where(kk < 0, (kk * (1 << _PRECISION_BITS) - 0.5), (kk * (1 << _PRECISION_BITS) + 0.5)).astype(np.int64).astype(np.int32)
    return xmin, cnt, ik, ksize


def cv_linear_taps(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """OpenCV resize INTER_LINEAR, 8-bit: (index0, index1, 11-bit weight of index1) per output coordinate."""
    pos = ((np.arange(out_size, dtype=np.float64) + 0.5) * (in_size / out_size) - 0.5).astype(np.float32)
    base = np.floor(pos).astype(np.int64)
    frac = pos - base.astype(np.float32)
    edge = (base < 0) | (base >= in_size - 1)
    base = np.clip(base, 0, in_size - 1)
    frac = np.where(edge, np.float32(0), frac)
    nxt = np.minimum(base + 1, in_size - 1)
    w1 = np.rint(frac * np.float32(2048)).astype(np.int32)
    return base.astype(np.int32), nxt.astype(np.int32), w1


def _tight(xmin, cnt, coef, ksize):
    """Pillow sizes its table rows for ceil(support) * 2 + 1 taps; the rows hold at most floor(2 * support) + 1 (250 -> 192:
    three of five).  The device kernels loop over the table WIDTH (zero weights included), so hand them the tight table."""
    k = max(1, int(cnt.max()))
    return xmin, cnt, np.ascontiguousarray(coef[:, :k]), k


@lru_cache(maxsize=64)
def _pil_tables_dev(H: int, W: int, oh: int, ow: int, device_index: int):
    torch = _lib.require_gpu()
    dev = torch.device("cuda", device_index)
    xm, xc, xk, xks = _tight(*pil_bilinear_coeffs(W, ow))
    ym, yc, yk, yks = _tight(*pil_bilinear_coeffs(H, oh))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return (t(xm), t(xc), t(xk), xks, t(ym), t(yc), t(yk), yks)


@lru_cache(maxsize=256)
def _cv_tables_dev(H: int, W: int, oh: int, ow: int, device_index: int):
    torch = _lib.require_gpu()
    dev = torch.device("cuda", device_index)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return tuple(t(a) for a in cv_linear_taps(W, ow)) + tuple(t(a) for a in cv_linear_taps(H, oh))


def _mean_array(imageNetUtilsMean: bool):
    m = IMAGENET_CAFFE_BGR_MEAN if imageNetUtilsMean else VGGFACE2_BGR_MEAN
    return (ctypes.c_double * 3)(float(m[0]), float(m[1]), float(m[2]))


def _as_u8_cuda(imgs, device=None):
    torch = _lib.require_gpu()
    if isinstance(imgs, np.ndarray):
        imgs = torch.from_numpy(np.ascontiguousarray(imgs, dtype=np.uint8)).to(_lib.cuda_device(device))
    elif device is not None and imgs.is_cuda and imgs.device != _lib.cuda_device(device):
        raise ValueError("images are on %s, expected %s" % (imgs.device, _lib.cuda_device(device)))
    if imgs.dtype != torch.uint8 or imgs.dim() != 4 or imgs.shape[3] != 3 or not imgs.is_cuda:
        raise ValueError("images must be a uint8 [n,H,W,3] RGB batch")
    return imgs.contiguous()


def preprocess_pil(imgs, out_hw: Tuple[int, int], convert2BGR: bool = True, imageNetUtilsMean: bool = True, device=None,
                   raw_u8: bool = False):
    """facerec_test.py:93-110 for a batch of same-size decoded RGB images -> CUDA float32 [n,oh,ow,3].
    A NumPy batch is uploaded to ``device`` (default: the current one); a CUDA batch stays where it is.
    raw_u8: stop after misc.imresize (:93) -- CUDA uint8 [n,oh,ow,3], RGB, for Engine.forward_u8, whose first kernel does the
    float conversion, channel reversal and mean subtraction (:95-106) inside its window load."""
    torch = _lib.require_gpu()
    x = _as_u8_cuda(imgs, device)
    n, H, W, _ = x.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    if raw_u8 and (H, W) == (oh, ow):
        return x                      # Pillow's resize to the same size is the identity (coefficients 1.0)
    xm, xc, xk, xks, ym, yc, yk, yks = _pil_tables_dev(H, W, oh, ow, x.device.index or 0)
    tmp = torch.empty((n, H, ow, 3), dtype=torch.uint8, device=x.device)
    out = torch.empty((n, oh, ow, 3), dtype=torch.uint8 if raw_u8 else torch.float32, device=x.device)
    mode = COLOR_NONE_U8 if raw_u8 else (COLOR_BGR_MEAN_F64 if convert2BGR else COLOR_RGB_UNIT)
    with _lib.on_device(x):
        _lib.check(_lib.lib().hsefr_preprocess_pil_u8(x.data_ptr(), tmp.data_ptr(), out.data_ptr(), n, H, W, oh, ow, xm.data_ptr(),
                                                      xc.data_ptr(), xk.data_ptr(), xks, ym.data_ptr(), yc.data_ptr(), yk.data_ptr(),
                                                      yks, mode, _mean_array(imageNetUtilsMean), _lib.current_stream_ptr()),
                   "hsefr_preprocess_pil_u8")
    return out


def preprocess_cv(imgs, out_hw: Tuple[int, int], device=None, raw_u8: bool = False):
    """facial_analysis.py:95-107 (cv2.resize -> float32 -> BGR -> ImageNet-Caffe mean) for a same-size batch.
    raw_u8: stop after cv2.resize (:95) -- CUDA uint8 [n,oh,ow,3] RGB for Engine.forward_u8."""
    torch = _lib.require_gpu()
    x = _as_u8_cuda(imgs, device)
    n, H, W, _ = x.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    if raw_u8 and (H, W) == (oh, ow):
        return x
    out = torch.empty((n, oh, ow, 3), dtype=torch.uint8 if raw_u8 else torch.float32, device=x.device)
    if (H, W) == (oh, ow):
        tabs = [None] * 6
    else:
        tabs = [t.data_ptr() for t in _cv_tables_dev(H, W, oh, ow, x.device.index or 0)]
    with _lib.on_device(x):
        _lib.check(_lib.lib().hsefr_preprocess_cv_u8(x.data_ptr(), out.data_ptr(), n, H, W, oh, ow, *tabs, COLOR_NONE_U8 if raw_u8 else COLOR_BGR_MEAN_F32,
                                                     _mean_array(True), _lib.current_stream_ptr()), "hsefr_preprocess_cv_u8")
    return out


def preprocess_faces_cv(crops: Sequence[np.ndarray], out_hw: Tuple[int, int], device=None, raw_u8: bool = False):
    """Variable-size face crops (process_image's per-box crops): grouped by size, one launch per group,
    results returned in input order as one CUDA float32 [n,oh,ow,3] tensor on ``device`` (raw_u8: the resized RGB bytes,
    uint8, for Engine.forward_u8)."""
    torch = _lib.require_gpu()
    oh, ow = out_hw
    dev = _lib.cuda_device(device)
    groups = {}
    for i, c in enumerate(crops):
        groups.setdefault(c.shape[:2], []).append(i)
    if len(groups) == 1:       # one face, or faces of one size: the resize kernel's own output, no scatter (two launches and an index upload less)
        return preprocess_cv(np.stack([np.ascontiguousarray(c, dtype=np.uint8) for c in crops]), out_hw, dev, raw_u8=raw_u8)
    out = torch.empty((len(crops), oh, ow, 3), dtype=torch.uint8 if raw_u8 else torch.float32, device=dev)
    for (h, w), idx in groups.items():
        batch = np.stack([np.ascontiguousarray(crops[i], dtype=np.uint8) for i in idx])
        out[torch.tensor(idx, device=dev)] = preprocess_cv(batch, out_hw, dev, raw_u8=raw_u8)
    return out
