"""Host-side image preparation of the reference, kept pluggable (L1 in SURVEY §1).

``scipy.misc.imread/imresize`` (what facerec_test.py:83-93 calls; removed from SciPy 1.3+) were
thin wrappers over PIL, so they are provided here through PIL directly.  ``cv2.resize`` is not
available on this image; its INTER_LINEAR 8-bit path is implemented in ``resize_linear_u8``.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

IMAGENET_CAFFE_BGR_MEAN = np.array([103.939, 116.779, 123.68], dtype=np.float64)     # facerec_test.py:99-102
VGGFACE2_BGR_MEAN = np.array([91.4953, 103.8827, 131.0912], dtype=np.float64)        # facerec_test.py:103-106


def imread_rgb(path: str) -> np.ndarray:
    """misc.imread(path, mode='RGB') (facerec_test.py:83,91)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def imresize_bilinear(img: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """misc.imresize(img, size, interp='bilinear') (facerec_test.py:84,93): uint8 in, uint8 out,
    PIL's antialiasing BILINEAR filter; ``size`` is (rows, cols)."""
    from PIL import Image
    rows, cols = int(size[0]), int(size[1])
    return np.asarray(Image.fromarray(np.asarray(img, dtype=np.uint8)).resize((cols, rows), resample=Image.BILINEAR))


def center_crop_250_128(img: np.ndarray) -> np.ndarray:
    """The crop_center branch, facerec_test.py:81-89: resize to 250x250, keep the middle 128x128."""
    orig_w, orig_h = 250, 250
    img = imresize_bilinear(img, (orig_w, orig_h))
    w1, h1 = 128, 128
    dw = (orig_w - w1) // 2
    dh = (orig_h - h1) // 2
    return img[dh:-dh, dw:-dw]


def to_model_input(resized_u8: np.ndarray, convert2BGR: bool, imageNetUtilsMean: bool, dtype=float) -> np.ndarray:
    """facerec_test.py:93-110 after the resize: float, optional RGB->BGR + mean, else x/127.5-1."""
    x = resized_u8.astype(dtype)
    if convert2BGR:
        x = x[..., ::-1].copy()
        mean = IMAGENET_CAFFE_BGR_MEAN if imageNetUtilsMean else VGGFACE2_BGR_MEAN
        # Python floats, as in the reference (`x[..., 0] -= 103.939`): with a float32 image the subtraction
        # is float32 arithmetic (a NumPy float64 scalar would silently promote it under NumPy >= 2)
        x[..., 0] -= float(mean[0])
        x[..., 1] -= float(mean[1])
        x[..., 2] -= float(mean[2])
    else:
        x /= 127.5
        x -= 1.0
    return x


def resize_linear_u8(img: np.ndarray, w: int, h: int) -> np.ndarray:
    """cv2.resize(img, (w, h)) with the default INTER_LINEAR on 8-bit images
    (facial_analysis.py:95): half-pixel centres, clamped borders, no antialiasing, 11-bit
    fixed-point tap weights and OpenCV's two-stage rounding."""
    img = np.asarray(img, dtype=np.uint8)
    ih, iw = img.shape[:2]
    if (ih, iw) == (h, w):
        return img.copy()

    def axis_taps(n_in: int, n_out: int):
        pos = ((np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5).astype(np.float32)
        base = np.floor(pos).astype(np.int64)
        frac = pos - base.astype(np.float32)
        under, over = base < 0, base >= n_in - 1
        base = np.clip(base, 0, n_in - 1)
        frac = np.where(under | over, np.float32(0), frac)
        nxt = np.minimum(base + 1, n_in - 1)
        b = np.rint(frac * np.float32(2048)).astype(np.int64)
        return base, nxt, 2048 - b, b

    r0, r1, a0, a1 = axis_taps(ih, h)
    c0, c1, b0, b1 = axis_taps(iw, w)
    src = img.astype(np.int64)
    if src.ndim == 2:
        src = src[..., None]
    top, bot = src[r0], src[r1]
    ht = top[:, c0] * b0[None, :, None] + top[:, c1] * b1[None, :, None]
    hb = bot[:, c0] * b0[None, :, None] + bot[:, c1] * b1[None, :, None]
    out = (((a0[:, None, None] * (ht >> 4)) >> 16) + ((a1[:, None, None] * (hb >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[..., 0] if img.ndim == 2 else out


# ---- cv2.resize(..., interpolation=cv2.INTER_AREA), used by the MTCNN cascade (facial_analysis.py:507,546,575) -------
def _area_weights(ssize: int, dsize: int):
    """Per destination index: source taps and coverage weights of OpenCV's decimation table
    (first partial pixel, fully covered pixels, last partial pixel), padded to a rectangular [dsize, T] table."""
    import math
    scale = ssize / dsize
    taps, wts = [], []
    for d in range(dsize):
        lo = d * scale
        hi = lo + scale
        cell = min(scale, ssize - lo)
        a, b = math.ceil(lo), min(math.floor(hi), ssize - 1)
        a = min(a, b)
        t, w = [], []
        if a - lo > 1e-3:
            t.append(a - 1); w.append((a - lo) / cell)
        for sx in range(a, b):
            t.append(sx); w.append(1.0 / cell)
        if hi - b > 1e-3:
            t.append(b); w.append(min(min(hi - b, 1.0), cell) / cell)
        taps.append(t); wts.append(w)
    T = max(len(t) for t in taps)
    idx = np.zeros((dsize, T), np.int64)
    wt = np.zeros((dsize, T), np.float64)
    for d, (t, w) in enumerate(zip(taps, wts)):
        idx[d, :len(t)] = t
        wt[d, :len(t)] = w
    return idx, wt


def _area_linear_taps(ssize: int, dsize: int):
    """INTER_AREA when enlarging: OpenCV's bilinear taps with the area-mode coordinate rule."""
    scale = ssize / dsize
    inv = 1.0 / scale
    d = np.arange(dsize)
    s0 = np.floor(d * scale).astype(np.int64)
    f = ((d + 1) - (s0 + 1) * inv).astype(np.float32)
    f = np.where(f <= 0, np.float32(0), f - np.floor(f)).astype(np.float32)
    edge = s0 >= ssize - 1
    s0 = np.where(edge, ssize - 1, s0)
    f = np.where(edge, np.float32(0), f)
    return s0, np.minimum(s0 + 1, ssize - 1), f


def resize_area(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv2.resize(src, (dw, dh), interpolation=cv2.INTER_AREA) for uint8 or float64 [H,W,C] images.
    Shrinking: box/coverage-weighted averaging accumulated tap by tap in OpenCV's order (float32 accumulators for
    8-bit images, float64 for float64 images); enlarging: the bilinear path with INTER_AREA's coordinates."""
    src = np.asarray(src)
    sh, sw = src.shape[:2]
    if (sh, sw) == (dh, dw):
        return src.copy()
    u8 = src.dtype == np.uint8
    acc_t = np.float32 if u8 else np.float64
    fx, fy = sw / dw, sh / dh
    if fx >= 1 and fy >= 1:
        ix, iy = int(round(fx)), int(round(fy))
        eps = np.finfo(np.float64).eps
        if abs(fx - ix) < eps and abs(fy - iy) < eps:            # exact integer factors: plain box filter
            blk = src.reshape(dh, iy, dw, ix, -1)
            if not u8:
                return blk.sum(axis=(1, 3)) * (1.0 / (ix * iy))
            tot = blk.astype(np.int64).sum(axis=(1, 3))
            if ix == 2 and iy == 2:
                return ((tot + 2) >> 2).astype(np.uint8)
            return np.clip(np.rint(tot.astype(np.float32) * np.float32(1.0 / (ix * iy))), 0, 255).astype(np.uint8)
        xi, xw = _area_weights(sw, dw)
        yi, yw = _area_weights(sh, dh)
        S = src.astype(acc_t)
        rows = np.zeros((sh, dw, S.shape[2]), acc_t)             # horizontal pass, taps in table order
        for t in range(xi.shape[1]):
            rows += S[:, xi[:, t], :] * xw[:, t].astype(acc_t)[None, :, None]
        out = None
        for t in range(yi.shape[1]):                              # vertical pass: first tap assigns, later taps add
            term = rows[yi[:, t]] * yw[:, t].astype(acc_t)[:, None, None]
            out = term if out is None else out + term
        return np.clip(np.rint(out), 0, 255).astype(np.uint8) if u8 else out
    x0, x1, ax = _area_linear_taps(sw, dw)
    y0, y1, ay = _area_linear_taps(sh, dh)
    if u8:
        a1 = np.rint(ax * np.float32(2048)).astype(np.int64)
        a0 = np.rint((np.float32(1) - ax) * np.float32(2048)).astype(np.int64)
        b1 = np.rint(ay * np.float32(2048)).astype(np.int64)
        b0 = np.rint((np.float32(1) - ay) * np.float32(2048)).astype(np.int64)
        s = src.astype(np.int64)
        top, bot = s[y0], s[y1]
        h0 = top[:, x0] * a0[None, :, None] + top[:, x1] * a1[None, :, None]
        h1 = bot[:, x0] * a0[None, :, None] + bot[:, x1] * a1[None, :, None]
        v = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
        return np.clip(v, 0, 255).astype(np.uint8)
    s = src.astype(np.float64)
    a1, a0 = ax.astype(np.float64), (np.float32(1) - ax).astype(np.float64)
    b1, b0 = ay.astype(np.float64), (np.float32(1) - ay).astype(np.float64)
    top, bot = s[y0], s[y1]
    h0 = top[:, x0] * a0[None, :, None] + top[:, x1] * a1[None, :, None]
    h1 = bot[:, x0] * a0[None, :, None] + bot[:, x1] * a1[None, :, None]
    return h0 * b0[:, None, None] + h1 * b1[:, None, None]
