"""CPU suite, part 1: the oracle itself -- against the committed golden vectors, against an
independent torch-CPU lowering of the same graph, and on TensorFlow-semantics corner cases."""
import os

import numpy as np
import pytest

from oracle import pipeline as opl
from oracle import tf_graph as tfo

from conftest import GOLDEN, MODEL_PB, TEST_IMAGE

FETCH = ["global_pooling/Mean:0", "age_pred/Softmax:0", "gender_pred/Sigmoid:0"]


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


def test_graph_structure_matches_reference_notebook():
    # AgeGenderIdentityDemo.ipynb:51-52 prints these tensor names/shapes for the age-gender graph
    nodes = tfo.load_graphdef(MODEL_PB)
    g = tfo.GraphOracle(nodes)
    assert len(nodes) == 379
    assert g.placeholder_shape("input_1:0") == [-1, 224, 224, 3]
    for t in FETCH:
        assert g.tensor_exists(t)
    ops = {}
    for n in nodes:
        ops[n.op] = ops.get(n.op, 0) + 1
    assert ops["Conv2D"] == 14 and ops["DepthwiseConv2dNative"] == 13 and ops["Dequantize"] == 30
    # constants pinned by the file itself
    bias = [n for n in nodes if n.name.startswith("gender_pred/bias")]
    assert any(abs(float(n.attr["value"].tensor.reshape(-1)[0]) - 0.11695335) < 1e-7 for n in bias if n.op == "Const")


def test_oracle_reproduces_golden_image_vectors():
    z = np.load(os.path.join(GOLDEN, "e2e_test_image.npz"))
    g = tfo.GraphOracle(MODEL_PB, np.float64)
    img = opl.imread_rgb(TEST_IMAGE)
    x = opl.preprocess_image(img, 192, 192, True, True)[None]
    f, a, ge = g.run(FETCH, {"input_1:0": x})
    assert rel(f[0], z["feat_192"]) < 1e-6 and rel(a[0], z["age_192"]) < 1e-6 and rel(ge[0], z["gender_192"]) < 1e-6
    # survey-time provisional values for the 224 case
    assert list(np.argsort(z["age_224"])[-2:]) == [35, 37]
    assert abs(float(z["gender_224"][0]) - 0.8217) < 2e-4
    assert abs(np.linalg.norm(z["feat_224"]) - 10.146) < 2e-3


def test_fp32_and_fp64_oracle_agree_within_the_parity_bar():
    z = np.load(os.path.join(GOLDEN, "e2e_synthetic.npz"))
    x = np.random.RandomState(123).uniform(-128, 128, (3, 96, 96, 3)).astype(np.float32)
    f32 = tfo.GraphOracle(MODEL_PB, np.float32).run(FETCH, {"input_1:0": x})
    assert rel(f32[0], z["feat_96"]) < 1e-5
    assert rel(f32[1], z["age_96"]) < 1e-5
    assert rel(f32[2], z["gender_96"]) < 1e-5


def test_numpy_restatement_matches_independent_torch_lowering():
    from oracle.torch_cpu import TorchGraphOracle
    z = np.load(os.path.join(GOLDEN, "e2e_synthetic.npz"))
    x = np.random.RandomState(123).uniform(-128, 128, (2, 100, 100, 3)).astype(np.float32)   # odd feature-map sizes
    got = TorchGraphOracle(MODEL_PB).run(FETCH + ["conv_dw_2_relu/clip_by_value:0"], {"input_1:0": x})
    assert rel(got[0], z["feat_100"]) < 1e-5
    assert rel(got[1], z["age_100"]) < 1e-5
    ref = tfo.GraphOracle(MODEL_PB, np.float64).run("conv_dw_2_relu/clip_by_value:0", {"input_1:0": x})
    assert got[3].shape == ref.shape == (2, 25, 25, 64)
    assert rel(got[3], ref) < 1e-5


@pytest.mark.parametrize("size,k,s,expect", [(224, 3, 2, (112, 0, 1)), (192, 3, 2, (96, 0, 1)), (7, 3, 2, (4, 1, 1)),
                                             (112, 3, 1, (112, 1, 1)), (6, 3, 1, (6, 1, 1)), (1, 3, 2, (1, 1, 1)),
                                             (224, 7, 2, (112, 2, 3)), (5, 1, 1, (5, 0, 0))])
def test_tf_same_padding_rule(size, k, s, expect):
    assert tfo.same_pad(size, k, s) == expect


def test_conv_ops_against_torch_on_odd_shapes():
    import torch
    import torch.nn.functional as F
    rs = np.random.RandomState(0)
    for (h, w, c, s) in [(9, 7, 8, 1), (10, 11, 4, 2), (7, 7, 16, 2)]:
        x = rs.randn(2, h, w, c)
        k = rs.randn(3, 3, c, 1)
        y = tfo.depthwise_conv2d(x, k, (s, s), "SAME")
        oh, pt, pb = tfo.same_pad(h, 3, s)
        ow, pl, pr = tfo.same_pad(w, 3, s)
        xt = F.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (pl, pr, pt, pb))
        yt = F.conv2d(xt, torch.from_numpy(k).permute(2, 3, 0, 1), stride=s, groups=c).permute(0, 2, 3, 1).numpy()
        assert y.shape == yt.shape == (2, oh, ow, c)
        assert rel(y, yt) < 1e-12
        kk = rs.randn(3, 3, c, 5)
        y2 = tfo.conv2d(x, kk, (s, s), "SAME")
        yt2 = F.conv2d(xt, torch.from_numpy(kk).permute(3, 2, 0, 1), stride=s).permute(0, 2, 3, 1).numpy()
        assert rel(y2, yt2) < 1e-12


def test_dequantize_min_first_rounds_the_range_minimum():
    q = np.arange(256, dtype=np.uint8)
    lo, hi = -0.7312, 1.913
    w = tfo.dequantize_min_first(q, lo, hi)
    step = (np.float32(hi) - np.float32(lo)) / 255
    assert w.dtype == np.float32
    assert np.allclose(np.diff(w), step, rtol=0, atol=5e-7)
    assert abs(w[0] / step - round(float(w[0] / step))) < 1e-3      # minimum snapped to a multiple of the step
    assert abs(w[0] - lo) <= step / 2 + 1e-7
    assert np.all(tfo.dequantize_min_first(q, 0.5, 0.5) == np.float32(0.5))


def test_age_decode_and_is_male_follow_the_reference():
    p = np.zeros(100, np.float32)
    p[30], p[31], p[5] = 0.5, 0.3, 0.2
    age, idx, norm = opl.decode_age(p)
    assert list(idx) == [30, 31]
    assert abs(age - (1 + 30 * 0.625 + 31 * 0.375)) < 1e-6       # facial_analysis.py:113-124
    assert opl.is_male(np.array([0.6])) and not opl.is_male(np.array([0.59]))


def test_cv2_style_resize_properties():
    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    assert np.array_equal(opl.cv2_resize_linear(img, 53, 37), img)                   # identity size
    flat = np.full((20, 30, 3), 77, np.uint8)
    assert np.all(opl.cv2_resize_linear(flat, 224, 224) == 77)                       # constants are preserved
    up = opl.cv2_resize_linear(img, 106, 74)
    assert up.shape == (74, 106, 3) and up.min() >= img.min() and up.max() <= img.max()
    # exact 2x upsample of a horizontal ramp: interior samples sit at 1/4, 3/4 between neighbours
    ramp = np.tile((np.arange(8) * 16).astype(np.uint8)[None, :, None], (4, 1, 3))
    r2 = opl.cv2_resize_linear(ramp, 16, 8)[0, :, 0].astype(int)
    assert list(r2[1:7]) == [4, 12, 20, 28, 36, 44]
