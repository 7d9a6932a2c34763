"""CPU suite, part 7: the host half of the device preprocessing (coefficient / tap tables), checked
against the REAL Pillow (what scipy.misc.imresize wrapped -- facerec_test.py:93) and the oracle's OpenCV
restatement by running the two integer passes in NumPy.  This leg IS pinned against the reference's library."""
import numpy as np
import pytest
from PIL import Image

from hse_facerec_tf_amd import preprocess_device as pd
from oracle import pipeline as opl


def emulate_pil(img, oh, ow):
    H, W = img.shape[:2]
    xm, xc, xk, _ = pd.pil_bilinear_coeffs(W, ow)
    ym, yc, yk, _ = pd.pil_bilinear_coeffs(H, oh)
    tmp = np.zeros((H, ow, 3), np.int64)
    for xx in range(ow):
        s = np.full((H, 3), 1 << 21, np.int64)
        for t in range(xc[xx]):
            s += img[:, xm[xx] + t, :].astype(np.int64) * int(xk[xx, t])
        tmp[:, xx, :] = np.clip(s >> 22, 0, 255)
    out = np.zeros((oh, ow, 3), np.int64)
    for yy in range(oh):
        s = np.full((ow, 3), 1 << 21, np.int64)
        for t in range(yc[yy]):
            s += tmp[ym[yy] + t] * int(yk[yy, t])
        out[yy] = np.clip(s >> 22, 0, 255)
    return out.astype(np.uint8)


@pytest.mark.parametrize("H,W,oh,ow", [(250, 250, 192, 192), (250, 250, 224, 224), (588, 784, 224, 224), (100, 80, 224, 224),
                                       (192, 192, 192, 192), (37, 53, 20, 31), (128, 128, 224, 224)])
def test_pil_tables_reproduce_pillow_bit_exactly(H, W, oh, ow):
    img = np.random.RandomState(H + W).randint(0, 256, (H, W, 3)).astype(np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(emulate_pil(img, oh, ow), ref)


def test_pil_tables_shape_and_normalisation():
    xm, xc, xk, ks = pd.pil_bilinear_coeffs(250, 192)
    assert ks == 5 and xk.shape == (192, 5) and xm.min() == 0 and (xm + xc).max() == 250
    sums = np.array([xk[i, :xc[i]].sum() for i in range(192)])
    assert np.abs(sums - (1 << 22)).max() <= 3                 # weights sum to 1.0 in 22-bit fixed point
    xm, xc, xk, ks = pd.pil_bilinear_coeffs(100, 224)           # upscale: support 1 -> at most 2 live taps
    assert ks == 3 and xc.max() <= 3


@pytest.mark.parametrize("n_in,n_out", [(250, 224), (37, 224), (500, 224), (224, 224), (7, 3)])
def test_cv_taps_match_oracle_restatement(n_in, n_out):
    rs = np.random.RandomState(n_in)
    img = rs.randint(0, 256, (n_in, n_in, 3)).astype(np.uint8)
    i0, i1, w1 = pd.cv_linear_taps(n_in, n_out)
    assert i0.min() >= 0 and i1.max() <= n_in - 1 and w1.min() >= 0 and w1.max() <= 2048
    src = img.astype(np.int64)
    a1, a0 = w1.astype(np.int64), 2048 - w1.astype(np.int64)
    h0 = src[i0][:, i0] * a0[None, :, None] + src[i0][:, i1] * a1[None, :, None]
    h1 = src[i1][:, i0] * a0[None, :, None] + src[i1][:, i1] * a1[None, :, None]
    out = (((a0[:, None, None] * (h0 >> 4)) >> 16) + ((a1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    assert np.array_equal(np.clip(out, 0, 255).astype(np.uint8), opl.cv2_resize_linear(img, n_out, n_out))
