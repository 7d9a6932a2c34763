"""Oracle part 2: the reference's Python-side steps around ``sess.run``, restated.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED vs TensorFlow.

Each function cites the reference lines it follows.  ``scipy.misc.imread/imresize``
(removed from SciPy >= 1.3) are restated through PIL, which is what they wrapped:
``imread(mode='RGB')`` = ``Image.open().convert('RGB')``; ``imresize(arr, size,
interp='bilinear')`` = ``Image.fromarray(uint8).resize((size[1], size[0]), BILINEAR)``.
``cv2.resize`` (INTER_LINEAR, half-pixel centres, no antialias) is restated in NumPy.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

from .tf_graph import GraphOracle

IMAGENET_BGR_MEAN = (103.939, 116.779, 123.68)      # facerec_test.py:99-102, facial_analysis.py:105-107
VGGFACE2_BGR_MEAN = (91.4953, 103.8827, 131.0912)   # facerec_test.py:103-106


# ---- facerec_test.py:80-112 ----------------------------------------------------
def imread_rgb(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def imresize_bilinear(img_u8: np.ndarray, size_hw: Tuple[int, int]) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.fromarray(img_u8).resize((size_hw[1], size_hw[0]), Image.BILINEAR))


def preprocess_image(img_u8: np.ndarray, w: int, h: int, convert2BGR: bool = True,
                     imageNetUtilsMean: bool = True, crop_center: bool = False) -> np.ndarray:
    """facerec_test.py:80-112 on an already decoded RGB uint8 image -> float64 [h?,w?,3].

    Note the reference passes ``(self.w, self.h)`` as imresize's (rows, cols) size
    (facerec_test.py:93); for the square inputs of every model in scope it is moot.
    """
    img = img_u8
    if crop_center:  # :81-89
        img = imresize_bilinear(img, (250, 250))
        dw = (250 - 128) // 2
        dh = (250 - 128) // 2
        img = img[dh:-dh, dw:-dw]
    x = imresize_bilinear(img, (w, h)).astype(float)  # :93
    if convert2BGR:  # :95-106
        x = x[..., ::-1].copy()
        mean = IMAGENET_BGR_MEAN if imageNetUtilsMean else VGGFACE2_BGR_MEAN
        x[..., 0] -= mean[0]
        x[..., 1] -= mean[1]
        x[..., 2] -= mean[2]
    else:  # :107-110
        x /= 127.5
        x -= 1.0
    return x


# ---- cv2.resize(img, (w, h)) default INTER_LINEAR ----------------------------------
def cv2_resize_linear(img_u8: np.ndarray, w: int, h: int) -> np.ndarray:
    """OpenCV INTER_LINEAR for uint8: half-pixel centres, edge clamp, no antialias,
    fixed-point weights with 11 fractional bits (INTER_RESIZE_COEF_BITS) and the
    final (+ (1<<21)) >> 22 rounding of the 8-bit path."""
    ih, iw = img_u8.shape[:2]
    if (ih, iw) == (h, w):
        return img_u8.copy()

    def taps(in_n: int, out_n: int):
        scale = in_n / out_n
        f = ((np.arange(out_n) + 0.5) * scale - 0.5).astype(np.float32)   # fx = (float)(...)
        i0 = np.floor(f).astype(np.int64)                                  # sx = cvFloor(fx)
        frac = (f - i0.astype(np.float32)).astype(np.float32)              # fx -= sx
        lo = i0 < 0
        i0[lo] = 0
        frac[lo] = 0.0
        hi = i0 >= in_n - 1
        i0[hi] = in_n - 1
        frac[hi] = 0.0
        i1 = np.minimum(i0 + 1, in_n - 1)
        w1 = np.rint(frac * 2048.0).astype(np.int64)   # cvRound -> saturate_cast<short>
        w0 = 2048 - w1
        return i0, i1, w0, w1

    y0, y1, wy0, wy1 = taps(ih, h)
    x0, x1, wx0, wx1 = taps(iw, w)
    src = img_u8.astype(np.int64)
    rows0 = src[y0]                 # [h, iw, c]
    rows1 = src[y1]
    h0 = rows0[:, x0] * wx0[None, :, None] + rows0[:, x1] * wx1[None, :, None]
    h1 = rows1[:, x0] * wx0[None, :, None] + rows1[:, x1] * wx1[None, :, None]
    # VResizeLinear<uchar,int,short>: ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2
    v = (((wy0[:, None, None] * (h0 >> 4)) >> 16) + ((wy1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


# ---- facial_analysis.py:93-129 ---------------------------------------------------
def age_gender_preprocess(img_rgb_u8: np.ndarray, w: int, h: int) -> np.ndarray:
    """facial_analysis.py:95-108 -> float32 [1,h,w,3] BGR, ImageNet-Caffe mean removed."""
    resized = cv2_resize_linear(img_rgb_u8, w, h)
    x = resized.astype(np.float32)
    x = x[..., ::-1].copy()
    x[..., 0] -= 103.939
    x[..., 1] -= 116.779
    x[..., 2] -= 123.68
    return x[None]


def decode_age(age_preds: np.ndarray, min_age: int = 1):
    """facial_analysis.py:112-124: expected age over the two most probable bins."""
    indices = age_preds.argsort()[::-1][:2]
    norm_preds = age_preds[indices] / np.sum(age_preds[indices])
    res_age = min_age
    for age, probab in zip(indices, norm_preds):
        res_age += age * probab
    return res_age, indices, norm_preds


def is_male(gender_preds) -> bool:
    """facial_analysis.py:76-81 (use_sota=False branch)."""
    return gender_preds >= 0.6


class OracleTensorFlowInference:
    """facerec_test.py:50-125 executed on the GraphOracle instead of tf.Session."""

    def __init__(self, frozen_graph_filename, input_tensor, output_tensor, learning_phase_tensor=None,
                 convert2BGR=True, imageNetUtilsMean=True, additional_input_value=0,
                 compute_dtype=np.float64, input_size: Optional[Tuple[int, int]] = None):
        self.g = GraphOracle(frozen_graph_filename, compute_dtype)
        for t in (input_tensor, output_tensor, learning_phase_tensor):
            if t is not None and not self.g.tensor_exists(t):
                raise KeyError("The name %r refers to a Tensor which does not exist." % t)
        self.input_tensor, self.output_tensor = input_tensor, output_tensor
        self.learning_phase_tensor = learning_phase_tensor
        shape = self.g.placeholder_shape(input_tensor)
        if shape is None:  # :66-67
            w = h = 160
        else:
            _, w, h, _ = shape
        if input_size is not None:
            w, h = input_size
        self.w, self.h = int(w), int(h)
        self.convert2BGR, self.imageNetUtilsMean = convert2BGR, imageNetUtilsMean
        self.additional_input_value = additional_input_value

    def preprocess_image(self, img_filepath, crop_center):
        return preprocess_image(imread_rgb(img_filepath), self.w, self.h, self.convert2BGR,
                                self.imageNetUtilsMean, crop_center)

    def extract_batch(self, x_nhwc: np.ndarray) -> np.ndarray:
        feed = {self.input_tensor: x_nhwc}
        if self.learning_phase_tensor is not None:
            feed[self.learning_phase_tensor] = self.additional_input_value
        out = self.g.run(self.output_tensor, feed)
        return np.asarray(out, np.float32).reshape(x_nhwc.shape[0], -1)

    def extract_features(self, img_filepath, crop_center=False):  # :114-122
        x = self.preprocess_image(img_filepath, crop_center)
        return self.extract_batch(np.expand_dims(x, 0)).reshape(-1)


class OracleAgeGender:
    """facial_analysis.py:83-130 (``load_age_gender``) on the GraphOracle."""

    def __init__(self, frozen_graph_filename, compute_dtype=np.float64, input_size=None):
        self.g = GraphOracle(frozen_graph_filename, compute_dtype)
        _, w, h, _ = self.g.placeholder_shape("input_1:0")
        if input_size is not None:
            w, h = input_size
        self.w, self.h = int(w), int(h)

    def run_batch(self, x: np.ndarray) -> Dict[str, np.ndarray]:
        age, gender, feats = self.g.run(
            ["age_pred/Softmax:0", "gender_pred/Sigmoid:0", "global_pooling/Mean:0"], {"input_1:0": x})
        return {"age_probs": np.asarray(age, np.float32), "gender": np.asarray(gender, np.float32),
                "features": np.asarray(feats, np.float32)}

    def age_gender_fun(self, img_rgb_u8: np.ndarray):
        r = self.run_batch(age_gender_preprocess(img_rgb_u8, self.w, self.h))
        res_age, _, _ = decode_age(r["age_probs"][0])
        return res_age, r["gender"][0], r["features"][0]
