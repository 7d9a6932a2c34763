"""The small-batch ("latency") plan: a second lowering of the same graph for the reference's one-image-per-run calls
(facerec_test.py:114-122, facial_analysis.py:93-129).  Chosen by the CALLER per call, never from the batch size alone: the bulk
paths keep one plan for every batch size, so an image's embedding does not depend on how the images were batched."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, MODEL_PB, TEST_IMAGE

pytestmark = pytest.mark.gpu
FETCH = ["global_pooling/Mean:0", "age_pred/Softmax:0", "gender_pred/Sigmoid:0"]
# north_star: embeddings within 1e-4 relative -- asserted per element by fp32_grade
from test_e2e_gpu import fp32_grade  # noqa: E402  (the element-wise form of the bar)


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def torch_():
    import torch
    assert torch.cuda.is_available()
    return torch


def _plans(size):
    from hse_facerec_tf_amd import graphdef, lowering
    g = graphdef.read_graph(MODEL_PB)
    outs = {0: FETCH[0], 1: FETCH[1], 2: FETCH[2]}
    big = lowering.lower_graph(g, "input_1:0", outs, (size, size), input_bound=256.0)
    small = lowering.lower_graph(g, "input_1:0", outs, (size, size), input_bound=256.0, presplit="none")
    return big, small


def test_routing_is_the_callers_choice_and_both_plans_meet_the_golden_bar(torch_):
    from hse_facerec_tf_amd import engine, lowering
    size = 192
    big, small = _plans(size)
    assert any(L.kind == lowering.OP_PWDW_PS for L in big.layers) and not any(L.kind in (lowering.OP_PWDW_PS, lowering.OP_PWGAP_PS) for L in small.layers)
    z = np.load(os.path.join(GOLDEN, "e2e_synthetic.npz"))
    n = z["feat_%d" % size].shape[0]
    x = torch_.from_numpy(np.random.RandomState(123).uniform(-128, 128, (n, size, size, 3)).astype(np.float32)).cuda()
    both = engine.Engine(big, max_batch=8, small_plan=small, small_batch=4)
    only_big, only_small = engine.Engine(big, max_batch=8), engine.Engine(small, max_batch=8)
    rb, rs = only_big.forward(x, (0, 1, 2)), only_small.forward(x, (0, 1, 2))
    for k, gk in (("features", "feat"), ("age_probs", "age"), ("gender", "gender")):
        fp32_grade(rb[k].cpu().numpy(), z["%s_%d" % (gk, size)], "bulk plan " + k)      # element-wise 1e-4 + max-norm 1e-5
        fp32_grade(rs[k].cpu().numpy(), z["%s_%d" % (gk, size)], "small plan " + k)
        assert rel(rs[k].cpu().numpy(), rb[k].cpu().numpy()) < 5e-6           # the two lowerings differ by summation order only
    assert n <= 4
    lat, bulk = both.forward(x, (0, 1, 2), latency=True), both.forward(x, (0, 1, 2))
    for k in ("features", "age_probs", "gender"):
        assert torch_.equal(lat[k], rs[k]), k          # latency=True, n <= small_batch: the small plan's bits
        assert torch_.equal(bulk[k], rb[k]), k         # default: the bulk plan's bits, whatever the batch size
    x5 = torch_.cat([x, x])[:5].contiguous()
    assert torch_.equal(both.forward(x5, (0,), latency=True)["features"], only_big.forward(x5, (0,))["features"])     # n > small_batch
    # every image's bulk embedding is independent of how it was batched
    one = torch_.cat([both.forward(x[i:i + 1].contiguous(), (0,))["features"] for i in range(n)])
    assert torch_.equal(one, rb["features"])
    assert both.device_bytes > only_big.device_bytes
    for e in (both, only_big, only_small):
        e.close()


def test_the_bound_flag_of_the_small_plan_is_seen(torch_):
    from hse_facerec_tf_amd import engine
    big, small = _plans(96)
    eng = engine.Engine(big, max_batch=4, small_plan=small)
    ok = torch_.zeros((1, 96, 96, 3), device="cuda")
    bad = ok.clone()
    bad[0, 5, 7, 1] = 300.0                         # outside the declared |x| < 256
    eng.forward(ok, latency=True)
    assert eng.input_overflow() is False
    eng.forward(bad, latency=True)
    assert eng.input_overflow() is True and eng.input_overflow() is False      # read-and-clear
    eng.forward(bad)
    assert eng.input_overflow() is True
    pinned = torch_.zeros(1, dtype=torch_.int32).pin_memory()
    eng.forward(bad, latency=True)
    eng.input_overflow_async(pinned.data_ptr())     # the flag of the engine the last forward ran on
    torch_.cuda.synchronize()
    assert int(pinned[0]) == 1
    eng.close()


def test_extractor_default_is_batching_invariant_and_latency_plan_is_opt_in(torch_):
    from hse_facerec_tf_amd import TensorFlowInference
    kw = dict(input_tensor="input_1:0", output_tensor=FETCH[0], convert2BGR=True, imageNetUtilsMean=True, input_size=(192, 192), max_batch=8)
    a, b = TensorFlowInference(MODEL_PB, **kw), TensorFlowInference(MODEL_PB, latency_plan=True, **kw)
    assert a.engine.small_plan is None and b.engine.small_plan is not None
    fa, fb = a.extract_features(TEST_IMAGE), b.extract_features(TEST_IMAGE)
    assert fa.shape == fb.shape == (1024,) and rel(fb, fa) < 5e-6 and not np.array_equal(fa, fb)
    bulk_a, bulk_b = a.extract_files([TEST_IMAGE] * 3, workers=1), b.extract_files([TEST_IMAGE] * 3, workers=1)
    assert np.array_equal(bulk_a, bulk_b)                       # the bulk path never takes the small plan
    assert all(np.array_equal(bulk_a[i], fa) for i in range(3))   # default: extract_files IS the loop of extract_features
    a.close_session()
    b.close_session()


def test_age_gender_fun_takes_the_latency_plan_by_default(torch_):
    from hse_facerec_tf_amd import FacialImageProcessing, preprocess
    rgb = preprocess.imread_rgb(TEST_IMAGE)
    face = np.ascontiguousarray(rgb[60:310, 250:500])
    fl = FacialImageProcessing(mtcnn_detector=False, max_batch=8)
    fb = FacialImageProcessing(mtcnn_detector=False, max_batch=8, latency_plan=False)
    assert fl.sess.small_plan is not None and fb.sess.small_plan is None
    al, gl, xl = fl.age_gender_fun(face)
    ab, gb, xb = fb.age_gender_fun(face)
    assert abs(al - ab) < 1e-3 and rel(gl, gb) < 1e-5 and rel(xl, xb) < 5e-6
    ages, genders, feats = fl.age_gender_batch([face, face])                # the bulk entry keeps the bulk plan
    assert np.array_equal(feats[0], xb) and np.array_equal(feats[1], xb) and ages[0] == ab
    ages_l, _, feats_l = fl.age_gender_batch([face], latency=True)
    assert np.array_equal(feats_l[0], xl) and ages_l[0] == al
    fl.close()
    fb.close()


def test_a_batch_of_5000_images_is_the_same_bits_as_batches_of_256(torch_):
    """A batch far beyond what the plan was tuned for goes through the network in chunks of engine.BULK_CHUNK images (consecutive
    layers meet in the caches; the workspace is one chunk's): every image's embedding equals the one a batch of 256 gives."""
    from hse_facerec_tf_amd import TensorFlowInference
    n = 5000
    tfi = TensorFlowInference(MODEL_PB, input_tensor="input_1:0", output_tensor=FETCH[0], input_size=(192, 192), max_batch=n)
    from hse_facerec_tf_amd.engine import BULK_CHUNK
    assert tfi.engine.chunk == BULK_CHUNK and tfi.engine.device_bytes < 2.5e9      # the workspace is one chunk's (5000 images in one piece: 12 GB)
    g = torch_.Generator(device="cuda")
    g.manual_seed(1)
    x = (torch_.rand((n, 192, 192, 3), device="cuda", generator=g) * 256 - 128).contiguous()
    big = tfi.engine.forward(x)["features"]
    for a in (0, 2400, 4700, n - 256):
        assert torch_.equal(tfi.engine.forward(x[a:a + 256].contiguous())["features"], big[a:a + 256]), a
    assert tfi.engine.input_overflow() is False
    tfi.close_session()


def test_chunked_and_unchunked_engines_agree_bit_for_bit_in_both_input_forms(torch_):
    """Engine(bulk_chunk=...) against bulk_chunk=0 (one piece; the streaming stem then splits its own launch beyond 2 GB of input),
    three outputs, float and uint8 entries, a ragged last chunk; more than max_batch still raises."""
    from hse_facerec_tf_amd import engine, graphdef, lowering
    g = graphdef.read_graph(MODEL_PB)
    outs = {0: FETCH[0], 1: FETCH[1], 2: FETCH[2]}
    plan = lowering.lower_graph(g, "input_1:0", outs, (96, 96), input_bound=256.0, u8_mean_bgr=(103.939, 116.779, 123.68))
    a, b = engine.Engine(plan, max_batch=300, bulk_chunk=64), engine.Engine(plan, max_batch=300, bulk_chunk=0)
    assert a.chunk == 64 and b.chunk == 300 and a.device_bytes < b.device_bytes
    gen = torch_.Generator(device="cuda")
    gen.manual_seed(9)
    x = (torch_.rand((203, 96, 96, 3), device="cuda", generator=gen) * 256 - 128).contiguous()
    u8 = torch_.randint(0, 256, (203, 96, 96, 3), dtype=torch_.uint8, device="cuda", generator=gen)
    ra, rb = a.forward(x, (0, 1, 2)), b.forward(x, (0, 1, 2))
    ua, ub = a.forward_u8(u8, (0, 2)), b.forward_u8(u8, (0, 2))
    for k in ("features", "age_probs", "gender"):
        assert torch_.equal(ra[k], rb[k]), k
    assert torch_.equal(ua["features"], ub["features"]) and torch_.equal(ua["gender"], ub["gender"])
    with pytest.raises(ValueError):
        a.forward(torch_.zeros((301, 96, 96, 3), device="cuda"))
    a.close()
    b.close()
