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
