"""Oracle part 5: ResNet-50 (`resnet50_ft`) forward, restated in NumPy with bf16 storage emulation.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the reference's file for this
network (models/vgg2_resnet.pb, facerec_test.py:213) is missing from the reference checkout, so
neither its exact graph nor any output of it exists; what is restated here is the published
VGGFace2/Caffe `resnet50_ft` topology that the tensor name `pool5_7x7_s1` belongs to,
written independently of hse_facerec_tf_amd/resnet50.py (explicit Python loops over stages, NumPy
convolutions from oracle/tf_graph.py).

Numerics contract being checked (BASELINE config 3: "bf16 MFMA"): weights and activations are
bf16 in memory, every convolution accumulates exactly (fp64 here, fp32 on the GPU), the folded
BatchNorm is applied in full precision, and each stored tensor is rounded to bf16 (round to nearest
even).  A bottleneck's last conv is rounded once before the shortcut is added and once after the
ReLU, as a layer-by-layer bf16 pipeline would do.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from . import tf_graph as tfo


def bf16_round(a: np.ndarray) -> np.ndarray:
    """Round float values to the nearest bf16 (ties to even), returned as float64."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def _conv_bn(x, w, name, stride, pad, rnd=bf16_round):
    k = rnd(w[name + "/kernel"])
    y = tfo.conv2d(x, k, (stride, stride), "", explicit_pads=(pad, pad, pad, pad))
    return y * w[name + "/scale"].astype(np.float64) + w[name + "/shift"].astype(np.float64)


def _maxpool_3x3_s2(x, mode):
    n, h, w, c = x.shape
    if mode == "caffe":   # pad 0, ceil mode: the last window may hang over the edge and is clipped
        oh, ow = -(-(h - 3) // 2) + 1, -(-(w - 3) // 2) + 1
    else:                 # 'valid'
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    ph, pw = max((oh - 1) * 2 + 3 - h, 0), max((ow - 1) * 2 + 3 - w, 0)
    xp = np.pad(x, ((0, 0), (0, ph), (0, pw), (0, 0)), constant_values=-np.inf)
    out = np.full((n, oh, ow, c), -np.inf)
    for dy in range(3):
        for dx in range(3):
            out = np.maximum(out, xp[:, dy:dy + 2 * (oh - 1) + 1:2, dx:dx + 2 * (ow - 1) + 1:2, :])
    return out


def forward(weights: Dict[str, np.ndarray], x_nhwc: np.ndarray, pool: str = "caffe", return_all: bool = False,
            storage: str = "bf16"):
    """x: [n,h,w,3] float (BGR, VGGFace2-mean-subtracted as facerec_test.py:103-106 leaves it) -> [n,2048].
    storage='bf16': the bf16 pipeline's contract (above); storage='exact': no rounding anywhere -- the fp64 network the
    reference's fp32 TensorFlow run approximates, the checker of the product's fp32-grade mode (1e-4 bar)."""
    acts = {}
    if storage == "exact":
        bf16_round = lambda a: np.asarray(a, np.float64)        # noqa: E731
    else:
        assert storage == "bf16"
        bf16_round = globals()["bf16_round"]
    _cb = _conv_bn
    _conv_bn_l = lambda x, w, name, stride, pad: _cb(x, w, name, stride, pad, bf16_round)   # noqa: E731
    x = bf16_round(x_nhwc)                                    # the stem kernel converts the image to bf16
    x = bf16_round(np.maximum(_conv_bn_l(x, weights, "conv1_7x7_s2", 2, 3), 0))
    acts["conv1_7x7_s2"] = x
    x = _maxpool_3x3_s2(x, pool)
    acts["pool1_3x3_s2"] = x
    plan = (("conv2", 3, 1), ("conv3", 4, 2), ("conv4", 6, 2), ("conv5", 3, 2))
    for stage, blocks, stride in plan:
        for b in range(1, blocks + 1):
            pre = "%s_%d" % (stage, b)
            s = stride if b == 1 else 1
            r = bf16_round(np.maximum(_conv_bn_l(x, weights, pre + "_1x1_reduce", s, 0), 0))
            t = bf16_round(np.maximum(_conv_bn_l(r, weights, pre + "_3x3", 1, 1), 0))
            shortcut = bf16_round(_conv_bn_l(x, weights, pre + "_1x1_proj", s, 0)) if b == 1 else x
            inc = bf16_round(_conv_bn_l(t, weights, pre + "_1x1_increase", 1, 0))
            x = bf16_round(np.maximum(inc + shortcut, 0))
            acts[pre + "_1x1_reduce"], acts[pre + "_3x3"], acts[pre + "_1x1_increase"] = r, t, x
            if b == 1:
                acts[pre + "_1x1_proj"] = shortcut
    feat = x.mean(axis=(1, 2))
    acts["pool5_7x7_s1"] = feat
    return (feat, acts) if return_all else feat
