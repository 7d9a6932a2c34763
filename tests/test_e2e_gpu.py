"""GPU parity, end to end: the engine behind the reference's API vs golden vectors and the live
oracle.  Bar (north_star): embeddings within 1e-4 relative, fp32 -- asserted ELEMENT-WISE by `fp32_grade` on every element
above 1e-2 of the tensor's maximum (1e-3 on those above 1e-3 of it), next to a max-norm bar of 1e-5 (`rel` is max|a-b| / max|b|;
observed 2e-6 .. 6e-6)."""
import os

import numpy as np
import pytest

from oracle import pipeline as opl
from oracle import tf_graph as tfo

from conftest import GOLDEN, MODEL_PB, TEST_IMAGE

pytestmark = pytest.mark.gpu

BAR = 1e-4            # north_star: within 1e-4 relative -- per element above 1e-2 of the maximum (fp32_grade)
BAR_MAXNORM = 1e-5    # max|a-b| / max|b| of the fp32-grade modes (a regression of one decade shows)
FETCH = ["global_pooling/Mean:0", "age_pred/Softmax:0", "gender_pred/Sigmoid:0"]


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


WORST = {"maxnorm": 0.0, "rel_above_1e-2": 0.0, "rel_above_1e-3": 0.0}      # observed over the session (printed by the last test of this file)


def fp32_grade(a, b, what=""):
    """The bar as BASELINE.json words it, asserted element by element: every element above 1e-2 of the tensor's maximum within 1e-4
    RELATIVE of the oracle's, every element above 1e-3 of it within 1e-3, and the whole tensor within 1e-5 of its scale.
    Why the relative bar steps with the element's size: an embedding element at 1e-3 of the maximum is a mean of sums whose terms are
    a thousand times larger; fp32 carries those terms to 6e-8 each, i.e. to ~1e-4 of THAT element -- the first version of this
    check (1e-4 from 1e-3 of the maximum up) failed at 1.6e-4 on one element while the vector stood at 2e-6 of its scale, and any
    fp32 evaluation of the graph, TensorFlow's included, would have.  Elements below 1e-3 of the maximum are covered by the max-norm."""
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a - b)
    WORST["maxnorm"] = max(WORST["maxnorm"], float(err.max() / scale))
    assert err.max() / scale < BAR_MAXNORM, "%s: max-norm error %.3g" % (what, err.max() / scale)
    for cut, bar, key in ((1e-2, BAR, "rel_above_1e-2"), (1e-3, 1e-3, "rel_above_1e-3")):
        big = np.abs(b) > cut * scale
        worst = float((err[big] / np.abs(b)[big]).max()) if big.any() else 0.0
        WORST[key] = max(WORST[key], worst)
        assert worst < bar, "%s: element-wise relative error %.3g on elements above %g of the maximum" % (what, worst, cut)


@pytest.fixture(scope="module")
def torch_():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_native_library_is_the_one_loaded(torch_):
    from hse_facerec_tf_amd import _lib
    _lib.lib()
    maps = open("/proc/self/maps").read()
    assert "hse_facerec_tf_amd/libhsefr.so" in maps


@pytest.mark.parametrize("stem_fusion", ["stem2", "stem", "none"])
@pytest.mark.parametrize("size", [192, 100])
def test_engine_matches_golden_for_every_stem_fusion(torch_, size, stem_fusion):
    from hse_facerec_tf_amd import _lib, engine, graphdef, lowering
    if stem_fusion == "stem" and not hasattr(_lib.lib(), "hsefr_stem_fused"):
        pytest.skip("HSEFR_OP_STEM_F16S (round 1's fused stem) runs on development builds of the library only")
    z = np.load(os.path.join(GOLDEN, "e2e_synthetic.npz"))
    n = z["feat_%d" % size].shape[0]
    plan = lowering.lower_graph(graphdef.read_graph(MODEL_PB), "input_1:0", {0: FETCH[0], 1: FETCH[1], 2: FETCH[2]}, (size, size),
                                stem_fusion=stem_fusion)
    want_kind = {"stem2": lowering.OP_STEM2_F16S, "stem": lowering.OP_STEM_F16S, "none": lowering.OP_CONV_C3}[stem_fusion]
    assert plan.layers[0].kind == want_kind
    eng = engine.Engine(plan, max_batch=4)
    x = np.random.RandomState(123).uniform(-128, 128, (n, size, size, 3)).astype(np.float32)
    out = eng.forward(torch_.from_numpy(x).cuda(), (0, 1, 2))
    fp32_grade(out["features"].cpu().numpy(), z["feat_%d" % size], "features %d" % size)
    fp32_grade(out["age_probs"].cpu().numpy(), z["age_%d" % size], "age_probs %d" % size)
    fp32_grade(out["gender"].cpu().numpy(), z["gender_%d" % size], "gender %d" % size)
    eng.close()


@pytest.mark.parametrize("size", [192, 100])
def test_block_fusion_is_bit_identical_to_the_unfused_plan(torch_, size):
    """block_fusion 'auto' (the HBM-bound stride-1 blocks) and 'all' (every block csrc/dwpw_f16s.hip covers, both
    strides, both kernel versions) against 'none': same golden parity, and the SAME BITS -- the fused kernels keep the
    operation order of the kernels they replace.  (With fp32 tensors between depthwise and pointwise layers: the pre-split
    LDS-DMA GEMM of round 2 sums a step's products in another order -- same bar, other bits -- so which layers take it must
    not differ between the plans compared bit for bit; the default plan is held to the golden bar below and in every
    other test of this file.)"""
    from hse_facerec_tf_amd import engine, graphdef, lowering
    z = np.load(os.path.join(GOLDEN, "e2e_synthetic.npz"))
    n = z["feat_%d" % size].shape[0]
    x = torch_.from_numpy(np.random.RandomState(123).uniform(-128, 128, (n, size, size, 3)).astype(np.float32)).cuda()
    outs = {}
    for mode in ("none", "auto", "all"):
        plan = lowering.lower_graph(graphdef.read_graph(MODEL_PB), "input_1:0", {0: FETCH[0], 1: FETCH[1], 2: FETCH[2]}, (size, size),
                                    block_fusion=mode, presplit="none")
        nfused = sum(L.kind == lowering.OP_DWPW_F16S for L in plan.layers)
        assert nfused == {"none": 0, "auto": 2, "all": 11}[mode]
        eng = engine.Engine(plan, max_batch=4)
        outs[mode] = {k: v.clone() for k, v in eng.forward(x, (0, 1, 2)).items()}
        eng.close()
        fp32_grade(outs[mode]["features"].cpu().numpy(), z["feat_%d" % size], "features %d" % size)
    for mode in ("auto", "all"):
        for k in outs["none"]:
            assert torch_.equal(outs[mode][k], outs["none"][k]), (mode, k)
    # the default plan (pre-split tensors where K >= 256): golden bar, and round-off away from the plans above
    plan = lowering.lower_graph(graphdef.read_graph(MODEL_PB), "input_1:0", {0: FETCH[0], 1: FETCH[1], 2: FETCH[2]}, (size, size))
    assert any(L.in_split for L in plan.layers)
    eng = engine.Engine(plan, max_batch=4)
    dflt = eng.forward(x, (0, 1, 2))
    fp32_grade(dflt["features"].cpu().numpy(), z["feat_%d" % size], "features %d" % size)
    for k in outs["none"]:
        assert rel(dflt[k].cpu().numpy(), outs["none"][k].cpu().numpy()) < 1e-5, k
    eng.close()


@pytest.mark.parametrize("pw_math", ["auto", "f32"])
@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("size", [192, 224, 96, 100])
def test_engine_matches_golden_synthetic(torch_, size, fuse, pw_math):
    from hse_facerec_tf_amd import engine, graphdef, lowering
    z = np.load(os.path.join(GOLDEN, "e2e_synthetic.npz"))
    n = z["feat_%d" % size].shape[0]
    plan = lowering.lower_graph(graphdef.read_graph(MODEL_PB), "input_1:0",
                                {0: FETCH[0], 1: FETCH[1], 2: FETCH[2]}, (size, size), fuse=fuse, pw_math=pw_math)
    assert any(L.a_log2 for L in plan.layers) == (pw_math == "auto")
    eng = engine.Engine(plan, max_batch=4)
    x = np.random.RandomState(123).uniform(-128, 128, (n, size, size, 3)).astype(np.float32)
    out = eng.forward(torch_.from_numpy(x).cuda(), (0, 1, 2))
    fp32_grade(out["features"].cpu().numpy(), z["feat_%d" % size], "features %d" % size)
    fp32_grade(out["age_probs"].cpu().numpy(), z["age_%d" % size], "age_probs %d" % size)
    fp32_grade(out["gender"].cpu().numpy(), z["gender_%d" % size], "gender %d" % size)
    eng.close()
    with pytest.raises(RuntimeError):
        eng.forward(torch_.from_numpy(x).cuda())


def test_small_batches_replay_a_captured_graph_bit_for_bit(torch_):
    """n <= 8 goes through hipGraph replay (engine-owned input copy): identical bits to plain launches, for every
    combination of requested outputs, changing input pointers and interleaved batch sizes."""
    from hse_facerec_tf_amd import engine, graphdef, lowering
    plan = lowering.lower_graph(graphdef.read_graph(MODEL_PB), "input_1:0", {0: FETCH[0], 1: FETCH[1], 2: FETCH[2]}, (96, 96))
    eng = engine.Engine(plan, max_batch=16)
    rs = np.random.RandomState(3)
    xs = [torch_.from_numpy(rs.uniform(-128, 128, (n, 96, 96, 3)).astype(np.float32)).cuda() for n in (1, 3, 8, 1, 9, 2)]
    eng.set_graph_batch(0)
    ref = [eng.forward(x, (0, 1, 2)) for x in xs]
    assert eng.graph_launches() == 0
    eng.set_graph_batch(8)
    for rep in range(2):
        for x, r in zip(xs, ref):
            out = eng.forward(x.clone(), (0, 1, 2))          # a fresh pointer every call
            for k in ("features", "age_probs", "gender"):
                assert torch_.equal(out[k], r[k]), (k, x.shape[0])
            only = eng.forward(x, (0,))
            assert torch_.equal(only["features"], r["features"])
    assert eng.graph_launches() == 2 * 2 * 5                  # the batch of 9 takes plain launches
    eng.close()


def test_every_layer_matches_the_oracle(torch_):
    """Layer-by-layer: each fused layer's output vs the graph tensor it stands for (unfused fp64
    oracle), through the per-kernel entry points with the plan's own weights."""
    from hse_facerec_tf_amd import graphdef, lowering, ops
    g = graphdef.read_graph(MODEL_PB)
    plan = lowering.lower_graph(g, "input_1:0", {0: FETCH[0], 1: FETCH[1], 2: FETCH[2]}, (96, 96))
    x = np.random.RandomState(9).uniform(-128, 128, (2, 96, 96, 3)).astype(np.float32)
    names = {li: nm for nm, li in plan.tensor_layer.items()}
    orc = tfo.GraphOracle(MODEL_PB, np.float64)
    want = orc.run([names[i] + ":0" for i in range(len(plan.layers))], {"input_1:0": x})
    d = lambda a: torch_.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    acts = {-1: d(x)}
    for i, L in enumerate(plan.layers):
        src = acts[L.src]
        if L.kind == lowering.OP_CONV_C3:
            y = ops.conv3x3_c3(src, d(L.w), d(L.shift), L.stride, L.act)
        elif L.kind == lowering.OP_DWCONV3X3:
            y = ops.dwconv3x3(src, d(L.w.reshape(3, 3, -1)), d(L.scale), d(L.shift), L.stride, L.act)
        elif L.kind == lowering.OP_PWCONV_F32 and L.a_log2 > 0:
            assert float(src.max()) <= 6.0 and float(src.min()) >= 0.0          # the bound the lowering relied on
            y = ops.pwconv1x1_f16split(src, L.w.reshape(L.w.shape[2], L.w.shape[3]).T, d(L.shift), L.act, L.a_log2)
        elif L.kind == lowering.OP_PWCONV_F32:
            y = ops.pwconv1x1(src, d(L.w.reshape(L.w.shape[2], L.w.shape[3]).T), d(L.shift), L.act)
        elif L.kind == lowering.OP_STEM2_F16S:
            y = ops.stem2_fused(src, d(L.w0), d(L.shift0), d(L.w.reshape(3, 3, -1)), d(L.scale), d(L.shift),
                                L.w2.reshape(L.w2.shape[2], L.w2.shape[3]).T, d(L.shift2), d(L.w3.reshape(3, 3, -1)), d(L.scale3),
                                d(L.shift3), L.act, L.a_log2)
        elif L.kind == lowering.OP_STEM_F16S:
            y = ops.stem_fused(src, d(L.w0), d(L.shift0), d(L.w.reshape(3, 3, -1)), d(L.scale), d(L.shift),
                               L.w2.reshape(L.w2.shape[2], L.w2.shape[3]).T, d(L.shift2), L.act, L.a_log2)
        elif L.kind == lowering.OP_DWPW_F32:
            y = ops.dwpw_fused(src, d(L.w.reshape(3, 3, -1)), d(L.scale), d(L.shift),
                               d(L.w2.reshape(L.w2.shape[2], L.w2.shape[3]).T), d(L.shift2), L.stride)
        elif L.kind == lowering.OP_DWPW_F16S:
            y = ops.dwpw_f16split(src, d(L.w.reshape(3, 3, -1)), d(L.scale), d(L.shift), L.w2.reshape(L.w2.shape[2], L.w2.shape[3]).T,
                                  d(L.shift2), L.stride, L.act, L.a_log2)
        elif L.kind == lowering.OP_PWDW_PS:
            assert float(src.max()) <= 6.0 and float(src.min()) >= 0.0
            ys = ops.pwconv1x1_presplit_dw(ops.split_rows_encode(src, L.a_log2), L.w.reshape(L.w.shape[2], L.w.shape[3]).T, d(L.shift),
                                           d(L.w3.reshape(3, 3, -1)), d(L.scale3), d(L.shift3), L.act, L.a_log2, L.out_split)
            y = ops.split_rows_decode(ys, L.out_split)
        elif L.kind == lowering.OP_PWGAP_PS:
            y = ops.pwconv1x1_presplit_gap(ops.split_rows_encode(src, L.a_log2), L.w.reshape(L.w.shape[2], L.w.shape[3]).T, d(L.shift), L.act,
                                           L.a_log2).reshape(src.shape[0], 1, 1, -1)
        elif L.kind == lowering.OP_GAP:
            y = ops.gap(src)
        elif L.kind == lowering.OP_DENSE:
            y = ops.dense(src.reshape(src.shape[0], -1), d(L.w), d(L.shift), L.act)
        elif L.kind == lowering.OP_SOFTMAX:
            y = ops.softmax(src.reshape(src.shape[0], -1))
        acts[i] = y
        w = np.asarray(want[i]).reshape(y.shape)
        # (max-norm only: a layer's small outputs are differences of large sums, the embedding's are means of 36 pixels)
        assert rel(y.cpu().numpy(), w) < BAR_MAXNORM, "layer %d (%s)" % (i, L.name)


def test_tensorflow_inference_dropin_on_the_reference_image(torch_):
    """facerec_test.py usage: construct by tensor names, extract_features(path) -> flat fp32 vector."""
    from hse_facerec_tf_amd import TensorFlowInference
    z = np.load(os.path.join(GOLDEN, "e2e_test_image.npz"))
    tfi = TensorFlowInference(MODEL_PB, input_tensor='input_1:0', output_tensor='global_pooling/Mean:0',
                              convert2BGR=True, imageNetUtilsMean=True)
    assert (tfi.w, tfi.h) == (224, 224)
    f = tfi.extract_features(TEST_IMAGE)
    assert f.shape == (1024,) and f.dtype == np.float32
    fp32_grade(f, z["feat_224"], "z['feat_224']")
    x = tfi.preprocess_image(TEST_IMAGE, False)
    assert x.shape == (224, 224, 3) and x.dtype == np.float64
    tfi.close_session()
    t192 = TensorFlowInference(MODEL_PB, 'input_1:0', 'global_pooling/Mean:0', input_size=(192, 192))
    fp32_grade(t192.extract_features(TEST_IMAGE), z["feat_192"], "z['feat_192']")
    fb = t192.extract_files([TEST_IMAGE, TEST_IMAGE, TEST_IMAGE], batch=2)
    assert fb.shape == (3, 1024)
    fp32_grade(fb[2], z["feat_192"], "extract_files")
    assert np.array_equal(fb[0], fb[1]) and np.array_equal(fb[0], fb[2])
    with pytest.raises(ValueError):            # wrong spatial size fed, like sess.run's shape check
        t192.extract_batch(torch_.zeros((1, 224, 224, 3), device="cuda"))
    with pytest.raises(ValueError):            # more than max_batch
        t192.engine.forward(torch_.zeros((257, 192, 192, 3), device="cuda"))
    t192.close_session()


def test_facial_image_processing_dropin(torch_):
    """facial_analysis.py usage: age_gender_fun(face_rgb) and process_image(frame_bgr) with boxes."""
    from hse_facerec_tf_amd import FacialImageProcessing
    z = np.load(os.path.join(GOLDEN, "e2e_test_image.npz"))
    fip = FacialImageProcessing(print_stat=False, mtcnn_detector=False)
    img = opl.imread_rgb(TEST_IMAGE)
    age, gender, feats = fip.age_gender_fun(img)
    assert abs(age - float(z["ag_res_age"])) < 1e-2
    fp32_grade(gender, z["ag_gender"], "gender")
    fp32_grade(feats, z["ag_feat"], "features")
    assert gender.shape == (1,) and feats.shape == (1024,)
    bgr = np.ascontiguousarray(img[..., ::-1])
    bboxes, points, ages, genders, ffs = fip.process_image(bgr, bounding_boxes=z["boxes"])
    assert len(bboxes) == 4 and bboxes[3][0] == 0
    fp32_grade(np.asarray(ffs), z["crop_feats"], "z['crop_feats']")
    fp32_grade(np.asarray(genders), z["crop_genders"], "z['crop_genders']")
    assert np.abs(np.asarray(ages) - z["crop_ages"]).max() < 1e-2
    with pytest.raises(NotImplementedError):
        fip.process_image(bgr)                 # mtcnn_detector=False and no detector injected
    fip.close()


def test_batch_256_properties_at_full_size(torch_):
    """BASELINE config 2 at full size (batch 256, 192x192x3), where the oracle is too slow:
    size-independent properties -- every image's embedding is independent of its batch position
    and neighbours (bit-exact), and a sample of rows matches the oracle."""
    from hse_facerec_tf_amd import TensorFlowInference
    tfi = TensorFlowInference(MODEL_PB, 'input_1:0', 'global_pooling/Mean:0', input_size=(192, 192), max_batch=256)
    rs = np.random.RandomState(123)
    x = torch_.from_numpy(rs.uniform(-128, 128, (256, 192, 192, 3)).astype(np.float32)).cuda()
    full = tfi.extract_batch(x)
    assert tuple(full.shape) == (256, 1024)
    assert bool(torch_.isfinite(full).all()) and float(full.min()) >= 0.0 and float(full.max()) <= 6.0
    perm = torch_.from_numpy(rs.permutation(256)).cuda()
    assert torch_.equal(tfi.extract_batch(x[perm].contiguous()), full[perm])          # permutation equivariance
    for lo, n in ((0, 1), (17, 3), (250, 6)):                                           # batch-size independence
        assert torch_.equal(tfi.extract_batch(x[lo:lo + n].contiguous()), full[lo:lo + n])
    again = tfi.extract_batch(x)
    assert torch_.equal(again, full)                                                    # deterministic
    rows = [0, 131, 255]
    ref = tfo.GraphOracle(MODEL_PB, np.float64).run(FETCH[0], {"input_1:0": x[rows].cpu().numpy()})
    fp32_grade(full[rows].cpu().numpy(), ref, "ref")
    tfi.close_session()


def test_batch_512_properties_at_full_size(torch_):
    """BASELINE config 4 at full size (age/gender multi-head MobileNet-224, batch 512, THREE outputs), where the oracle is
    too slow: every image's features / age distribution / gender are independent of its batch position and neighbours
    (bit-exact), the softmax rows sum to one, and a sample of rows matches the oracle at the 1e-4 bar."""
    from hse_facerec_tf_amd import graphdef, lowering
    from hse_facerec_tf_amd.engine import Engine
    plan = lowering.lower_graph(graphdef.read_graph(MODEL_PB), "input_1:0", {0: FETCH[0], 1: FETCH[1], 2: FETCH[2]})
    assert plan.in_hwc == (224, 224, 3)
    eng = Engine(plan, max_batch=512)
    g = torch_.Generator(device="cuda").manual_seed(4)
    x = (torch_.rand((512, 224, 224, 3), device="cuda", generator=g) * 256.0 - 128.0).contiguous()
    full = {k: v.clone() for k, v in eng.forward(x, (0, 1, 2)).items()}
    assert tuple(full["features"].shape) == (512, 1024) and tuple(full["age_probs"].shape) == (512, 100) and tuple(full["gender"].shape) == (512, 1)
    assert all(bool(torch_.isfinite(v).all()) for v in full.values())
    assert float((full["age_probs"].sum(dim=1) - 1).abs().max()) < 1e-5
    assert 0.0 <= float(full["gender"].min()) and float(full["gender"].max()) <= 1.0
    perm = torch_.randperm(512, device="cuda", generator=g)
    again = eng.forward(x[perm].contiguous(), (0, 1, 2))
    for k in full:
        assert torch_.equal(again[k], full[k][perm]), k                                  # permutation equivariance
    for lo, n in ((0, 1), (300, 5), (505, 7)):                                           # batch-size independence
        part = eng.forward(x[lo:lo + n].contiguous(), (0, 1, 2))
        for k in full:
            assert torch_.equal(part[k], full[k][lo:lo + n]), (k, lo)
    rows = [0, 257, 511]
    ref = tfo.GraphOracle(MODEL_PB, np.float64).run(list(FETCH), {"input_1:0": x[rows].cpu().numpy()})
    for k, r in zip(("features", "age_probs", "gender"), ref):
        fp32_grade(full[k][rows].cpu().numpy(), r.reshape(3, -1), k)
    eng.close()


def test_vgg2_mobilenet_shaped_graph_through_the_reference_registry(torch_, tmp_path):
    """facerec_test.py:212 exactly: TensorFlowInference('models/vgg2_mobilenet.pb', input_tensor='input_1:0',
    output_tensor='reshape_1/Reshape:0', learning_phase_tensor='conv1_bn/keras_learning_phase:0', convert2BGR=True,
    imageNetUtilsMean=True) on a graph of that file's shape (tests/keras_mobilenet_graph.py: all 13 blocks, un-folded
    BatchNormalization behind learning-phase Switch/Merge, Relu6, reshape_1) at 192x192 -- against the unfused oracle at the
    1e-4 bar and against the shipped folded trunk."""
    import keras_mobilenet_graph as kg
    from hse_facerec_tf_amd import get_tf_face_recognizer, TensorFlowInference
    data = kg.build(MODEL_PB, 192)
    (tmp_path / "vgg2_mobilenet.pb").write_bytes(data)
    tfi = get_tf_face_recognizer("vgg2_mobilenet", models_dir=str(tmp_path), max_batch=8)
    assert (tfi.w, tfi.h) == (192, 192) and tfi.feature_dim == 1024 and tfi.tf_learning_phase == 'conv1_bn/keras_learning_phase:0'
    x = np.random.RandomState(212).uniform(-128, 128, (3, 192, 192, 3)).astype(np.float32)
    got = tfi.extract_batch(x)
    want = tfo.GraphOracle(tfo.parse_graphdef(data), np.float64).run(
        "reshape_1/Reshape:0", {"input_1:0": x[:2], "conv1_bn/keras_learning_phase:0": 0}).reshape(2, -1)
    fp32_grade(got[:2], want, "vgg2_mobilenet-shaped graph")
    folded = TensorFlowInference(MODEL_PB, 'input_1:0', 'global_pooling/Mean:0', input_size=(192, 192), max_batch=8)
    assert rel(got, folded.extract_batch(x)) < 1e-5
    # the file-path API of the reference on its demo image (preprocess_image + one run, facerec_test.py:114-122)
    f1 = tfi.extract_features(TEST_IMAGE)
    assert f1.shape == (1024,) and rel(f1, folded.extract_features(TEST_IMAGE)) < 1e-5
    tfi.close_session(), folded.close_session()


def test_vgg2_mobilenet_keras_h5_through_the_registry(torch_, tmp_path):
    """facerec_test.py:322-334 (model.load_weights('models/vgg2_mobilenet.h5'), output of 'reshape_1') without Keras or h5py:
    a weight file of that layout (tests/h5_writer.py: what save_weights writes) holding the shipped trunk un-folded into Keras
    layers -> get_tf_face_recognizer('vgg2_mobilenet_h5') -> the same embeddings as the shipped folded graph, on the GPU."""
    import h5_writer
    from test_h5weights_cpu import _keras_layers_of_the_shipped_trunk
    from hse_facerec_tf_amd import get_tf_face_recognizer, TensorFlowInference
    (tmp_path / "vgg2_mobilenet.h5").write_bytes(h5_writer.keras_save_weights(_keras_layers_of_the_shipped_trunk()))
    tfi = get_tf_face_recognizer("vgg2_mobilenet_h5", models_dir=str(tmp_path), max_batch=8)
    assert (tfi.w, tfi.h) == (192, 192) and tfi.feature_dim == 1024 and tfi.engine.accepts_u8
    x = np.random.RandomState(333).uniform(-128, 128, (3, 192, 192, 3)).astype(np.float32)
    folded = TensorFlowInference(MODEL_PB, 'input_1:0', 'global_pooling/Mean:0', input_size=(192, 192), max_batch=8)
    assert rel(tfi.extract_batch(x), folded.extract_batch(x)) < 2e-5
    assert rel(tfi.extract_features(TEST_IMAGE), folded.extract_features(TEST_IMAGE)) < 2e-5
    tfi.close_session(), folded.close_session()


@pytest.mark.parametrize("size,n_pwdw", [(192, 7), (224, 7)])
def test_epilogue_fused_plan_is_deterministic_and_equals_the_unfused_plan(torch_, size, n_pwdw):
    """The depthwise / pool epilogues of the pre-split GEMMs (lowering.fuse_pwdw / fuse_pwgap, on by default): 40 forwards of one
    batch are bit-identical (the epilogue hands tiles between waves through LDS: a missing barrier would show up as run-to-run
    noise), every batch size gives the rows of the full batch, and the features equal the unfused plan's to rounding.  Both
    plans: 192 x 192 (12- and 6-pixel maps: activations-first MFMA operands, dword parking) and 224 x 224 (14- and 7-pixel maps in
    tiles that advance by whole maps, the original operand order) -- VERDICT r2 weak 13."""
    from hse_facerec_tf_amd import graphdef, lowering
    from hse_facerec_tf_amd.engine import Engine
    g = graphdef.read_graph(MODEL_PB)
    fused = lowering.lower_graph(g, "input_1:0", {0: FETCH[0]}, (size, size), input_bound=256.0)
    plain = lowering.lower_graph(g, "input_1:0", {0: FETCH[0]}, (size, size), input_bound=256.0, pwdw_fusion="none")
    kinds = [L.kind for L in fused.layers]
    assert kinds.count(lowering.OP_PWDW_PS) == n_pwdw and kinds.count(lowering.OP_PWGAP_PS) == 1 and lowering.OP_PWDW_PS not in [L.kind for L in plain.layers]
    x = torch_.from_numpy(np.random.RandomState(21).uniform(-128, 128, (37, size, size, 3)).astype(np.float32)).cuda()
    ef, ep = Engine(fused, max_batch=37), Engine(plain, max_batch=37)
    ref = ef.forward(x)["features"].clone()
    for _ in range(40):
        assert torch_.equal(ef.forward(x)["features"], ref)
    for n in (1, 2, 3, 8, 36):
        assert torch_.equal(ef.forward(x[:n].contiguous())["features"], ref[:n])
    want = ep.forward(x)["features"]
    assert float((ref - want).abs().max() / want.abs().max()) < 2e-6
    ef.close(), ep.close()


def test_zz_report_the_worst_errors_the_bar_saw():
    """(runs last in this file) the largest errors fp32_grade met: visible with -s, and a guard that the bars are not vacuous."""
    print("fp32_grade over this session:", WORST)
    assert WORST["maxnorm"] < BAR_MAXNORM and WORST["rel_above_1e-2"] < BAR


def test_fused_heads_plan_equals_the_four_launch_plan(torch_):
    """lowering.mark_heads (round 6): the feats op carries HSEFR_OPF_HEADS and the engine runs feats + age_pred + softmax + gender_pred as
    one launch -- the features equal the plan lowered with launch_fusion=False bit for bit, age distribution, gender and the heads'
    four tensors to fp32 round-off (batch 37: a ragged last row group), and the three covered ops' profiled intervals are empty.  A forward that asks for the
    features alone runs none of the heads, as before."""
    from hse_facerec_tf_amd import graphdef, lowering
    from hse_facerec_tf_amd.engine import Engine
    g = graphdef.read_graph(MODEL_PB)
    outs = {0: FETCH[0], 1: FETCH[1], 2: FETCH[2]}
    fused = lowering.lower_graph(g, "input_1:0", outs, (96, 96), input_bound=256.0)
    plain = lowering.lower_graph(g, "input_1:0", outs, (96, 96), input_bound=256.0, launch_fusion=False)
    hi = [i for i, L in enumerate(fused.layers) if L.flags & lowering.OPF_HEADS]
    assert len(hi) == 1 and fused.layers[hi[0]].name.startswith("feats") and not any(L.flags for L in plain.layers)
    assert [L.kind for L in fused.layers[hi[0]:hi[0] + 4]] == [lowering.OP_DENSE, lowering.OP_DENSE, lowering.OP_SOFTMAX, lowering.OP_DENSE]
    ea, eb = Engine(fused, max_batch=37), Engine(plain, max_batch=37)
    gen = torch_.Generator(device="cuda").manual_seed(9)
    x = (torch_.rand((37, 96, 96, 3), device="cuda", generator=gen) * 256.0 - 128.0).contiguous()
    ra, rb = ea.forward(x, (0, 1, 2)), eb.forward(x, (0, 1, 2))
    assert torch_.equal(ra["features"], rb["features"])
    for k in ("age_probs", "gender"):          # (the fused heads split k sixteen ways, dense_kernel four: round-off apart)
        assert float((ra[k] - rb[k]).abs().max()) < 2e-6, k
    assert torch_.equal(ea.forward(x, (0,))["features"], rb["features"])
    assert torch_.equal(ea.forward(x, (2,))["gender"], rb["gender"])          # gender alone: age_pred / softmax are not needed -> four-launch path
    ea.forward_all_layers(x)
    eb.forward_all_layers(x)
    for i in range(hi[0], hi[0] + 4):
        ta, tb = ea.layer_output(i, 37), eb.layer_output(i, 37)
        assert float((ta - tb).abs().max()) <= 2e-6 * max(1.0, float(tb.abs().max())), fused.layers[i].name
    ea.set_profiling(1)
    ea.forward(x, (0, 1, 2))
    t = ea.op_times_ms(0)
    assert t[hi[0]] > 0 and max(t[hi[0] + 1:hi[0] + 4]) < 0.015      # (event records back to back: a few microseconds each)
    ea.close()
    eb.close()
