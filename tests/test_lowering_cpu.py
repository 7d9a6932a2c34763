"""CPU suite, part 2: the product's TF-free reader + lowering, executed on CPU by tests/plan_ref.py
(oracle ops over the SERIALISED plan) and compared with the oracle's unfused graph interpreter."""
import os
import struct

import numpy as np
import pytest

import plan_ref
from hse_facerec_tf_amd import graphdef, lowering
from hse_facerec_tf_amd.lowering import OUT_AGE, OUT_FEATURES, OUT_GENDER
from oracle import tf_graph as tfo

from conftest import GOLDEN, MODEL_PB

ALL_OUTS = {OUT_FEATURES: "global_pooling/Mean:0", OUT_AGE: "age_pred/Softmax:0", OUT_GENDER: "gender_pred/Sigmoid:0"}


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def graph():
    return graphdef.read_graph(MODEL_PB)


@pytest.fixture(autouse=True)
def _structure_tests_without_pwdw(monkeypatch):
    """The layer-table tests below count depthwise / pointwise layers of the passes they are about; the pointwise -> depthwise
    epilogue fusion (fuse_pwdw, on by default) has its own test, which asks for it explicitly."""
    real = lowering.lower_graph

    def without_pwdw(*args, **kwargs):
        kwargs.setdefault("pwdw_fusion", "none")
        return real(*args, **kwargs)
    monkeypatch.setattr(lowering, "lower_graph", without_pwdw)


def test_product_reader_agrees_with_oracle_reader(graph):
    nodes = tfo.load_graphdef(MODEL_PB)
    assert [n.name for n in nodes] == [n.name for n in graph.nodes]
    assert [n.op for n in nodes] == [n.op for n in graph.nodes]
    assert [n.inputs for n in nodes] == [n.inputs for n in graph.nodes]
    for a, b in zip(nodes, graph.nodes):
        if a.op == "Const":
            va, vb = a.attr["value"].tensor, graph.const_value(b)
            assert va.dtype == vb.dtype and va.shape == vb.shape and np.array_equal(va, vb)
    assert graph.placeholder_shape("input_1") == [-1, 224, 224, 3]


def test_get_tensor_by_name_error_behaviour(graph):
    with pytest.raises(KeyError):
        graph.get_tensor_by_name("no_such_op:0")
    with pytest.raises(ValueError):
        graph.get_tensor_by_name("input_1")            # op name, not a tensor name
    node, idx = graph.get_tensor_by_name("global_pooling/Mean:0")
    assert node.op == "Mean" and idx == 0


def test_plan_layer_table_matches_survey(graph):
    plan = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192), fuse=False)
    kinds = [L.kind for L in plan.layers]
    assert kinds[0] == lowering.OP_CONV_C3 and kinds.count(lowering.OP_DWCONV3X3) == 13
    assert kinds.count(lowering.OP_PWCONV_F32) == 13 and kinds.count(lowering.OP_GAP) == 1
    dw = [L for L in plan.layers if L.kind == lowering.OP_DWCONV3X3]
    assert [L.stride for L in dw] == [1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1]
    assert [L.in_shape[2] for L in dw] == [32, 64, 128, 128, 256, 256, 512, 512, 512, 512, 512, 512, 1024]
    assert all((L.pad_t, L.pad_l) == ((1, 1) if L.stride == 1 else (0, 0)) for L in dw)     # TF SAME, even inputs
    assert all(L.act == lowering.ACT_RELU6 for L in plan.layers[:27])
    # SURVEY 8d per-face figures for MobileNet-192
    assert abs(plan.bytes_per_image([lowering.OP_DWCONV3X3]) / 1e6 - 14.672) < 0.01
    assert abs(plan.bytes_per_image(range(1, 5)) / 1e6 - 30.085) < 0.01
    assert abs(plan.flops_per_image([lowering.OP_PWCONV_F32]) / 1e6 - 792.7) < 0.5
    plan224 = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, fuse=False)
    assert plan224.in_hwc == (224, 224, 3)
    assert abs(plan224.bytes_per_image([lowering.OP_DWCONV3X3]) / 1e6 - 19.970) < 0.01


@pytest.mark.parametrize("size,n", [(96, 3), (100, 2)])
def test_lowered_plan_equals_unfused_graph(graph, size, n):
    z = np.load(os.path.join(GOLDEN, "e2e_synthetic.npz"))
    plan = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (size, size))
    x = np.random.RandomState(123).uniform(-128, 128, (n, size, size, 3)).astype(np.float32)
    out = plan_ref.run(plan.serialize(), x)
    assert rel(out["features"], z["feat_%d" % size]) < 1e-6
    assert rel(out["age_probs"], z["age_%d" % size]) < 1e-6
    assert rel(out["gender"], z["gender_%d" % size]) < 1e-6


def test_fusion_pass_merges_the_early_blocks_only(graph):
    plan = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192), fuse_stem_block=False, block_fusion="none")
    kinds = [L.kind for L in plan.layers]
    assert kinds[:3] == [lowering.OP_CONV_C3, lowering.OP_DWPW_F32, lowering.OP_DWPW_F32]      # C = 32 and C = 64 blocks
    assert kinds.count(lowering.OP_DWPW_F32) == 2 and kinds.count(lowering.OP_DWCONV3X3) == 11
    f1, f2 = plan.layers[1], plan.layers[2]
    assert f1.in_shape == (96, 96, 32) and f1.out_shape == (96, 96, 64) and f1.stride == 1
    assert f2.in_shape == (96, 96, 64) and f2.out_shape == (48, 48, 128) and f2.stride == 2 and (f2.pad_t, f2.pad_l) == (0, 0)
    assert "conv_dw_1_relu/clip_by_value" not in plan.tensor_layer and plan.tensor_layer["conv_pw_1_relu/clip_by_value"] == 1
    # same FLOPs, 5.9 MB less HBM traffic per face (the two depthwise outputs are neither written nor re-read)
    unfused = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192), fuse=False, block_fusion="none")
    assert plan.flops_per_image() == unfused.flops_per_image()
    saved = unfused.bytes_per_image() - plan.bytes_per_image()
    assert saved == 2 * 4 * (96 * 96 * 32 + 48 * 48 * 64)
    # a depthwise tensor that is itself requested is not fused away
    keep = lowering.lower_graph(graph, "input_1:0", {OUT_FEATURES: "conv_dw_1_relu/clip_by_value:0"}, (64, 64), block_fusion="none")
    assert keep.layers[-1].kind == lowering.OP_DWCONV3X3
    # default: conv1 joins the first block (one kernel for graph nodes #30-#49), another 2.36 MB per face never reach HBM
    stem = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192), stem_fusion="stem", block_fusion="none")
    ks = [L.kind for L in stem.layers]
    assert ks[:3] == [lowering.OP_STEM_F16S, lowering.OP_DWPW_F32, lowering.OP_DWCONV3X3] and len(ks) == len(kinds) - 1
    s0 = stem.layers[0]
    assert s0.src == -1 and s0.in_shape == (192, 192, 3) and s0.out_shape == (96, 96, 64) and s0.stride == 2 and s0.a_log2 == 12
    assert stem.layers[1].src == 0 and stem.flops_per_image() == unfused.flops_per_image()
    assert plan.bytes_per_image() - stem.bytes_per_image() == 2 * 4 * 96 * 96 * 32
    assert "conv1_relu/clip_by_value" not in stem.tensor_layer and stem.tensor_layer["conv_pw_1_relu/clip_by_value"] == 0
    import plan_ref
    assert plan_ref.parse(stem.serialize())["ops"][0][0] == lowering.OP_STEM_F16S
    # default: the stride-2 depthwise of block 2 joins as well (graph nodes #30-#55); block 2's pointwise is a plain GEMM then
    s2 = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192), block_fusion="none")
    k2 = [L.kind for L in s2.layers]
    assert k2[:4] == [lowering.OP_STEM2_F16S, lowering.OP_PWCONV_F32, lowering.OP_DWCONV3X3, lowering.OP_PWCONV_F32]
    assert lowering.OP_DWPW_F32 not in k2 and len(k2) == len(kinds) - 1
    L0 = s2.layers[0]
    assert L0.in_shape == (192, 192, 3) and L0.out_shape == (48, 48, 64) and L0.pad3 == (0, 0) and L0.act == lowering.ACT_RELU6
    assert s2.layers[1].src == 0 and s2.layers[1].a_log2 == 12 and s2.layers[1].out_shape == (48, 48, 128)
    assert s2.flops_per_image() == unfused.flops_per_image()
    assert unfused.bytes_per_image() - s2.bytes_per_image() == 2 * 4 * (96 * 96 * 32 * 2 + 96 * 96 * 64)
    assert s2.tensor_layer["conv_dw_2_relu/clip_by_value"] == 0 and "conv_pw_1_relu/clip_by_value" not in s2.tensor_layer
    assert plan_ref.parse(s2.serialize())["ops"][0][0] == lowering.OP_STEM2_F16S
    odd = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (100, 100), block_fusion="none")          # 100 -> 50 -> 25: even map, no top/left pad
    assert odd.layers[0].kind == lowering.OP_STEM2_F16S and odd.layers[0].pad3 == (0, 0) and odd.layers[0].out_shape == (25, 25, 64)
    odd2 = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (98, 98), block_fusion="none")           # 98 -> 49 -> 25: odd map pads one row on top/left
    assert odd2.layers[0].pad3 == (1, 1) and odd2.layers[0].out_shape == (25, 25, 64)
    with pytest.raises(ValueError):
        lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (64, 64), stem_fusion="all", block_fusion="none")
    # conv1's own tensor requested, or the fp32-only arithmetic: no stem fusion
    keep1 = lowering.lower_graph(graph, "input_1:0", {OUT_FEATURES: "global_pooling/Mean:0", OUT_AGE: "conv1_relu/clip_by_value:0"}, (64, 64), block_fusion="none")
    assert keep1.layers[0].kind == lowering.OP_CONV_C3
    assert lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (64, 64), pw_math="f32", block_fusion="none").layers[0].kind == lowering.OP_CONV_C3


def test_block_fusion_pass(graph):
    """The HBM-bound stride-1 blocks (128 -> 128 and 256 -> 256 channels) become ONE split-f16 fused layer each by default;
    'all' fuses every depthwise -> pointwise pair the kernel covers; the serialised plan still evaluates to the same graph."""
    base = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192), block_fusion="none")
    auto = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192))
    every = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192), block_fusion="all")
    fused = [L for L in auto.layers if L.kind == lowering.OP_DWPW_F16S]
    assert [(L.in_shape, L.out_shape, L.stride, L.a_log2) for L in fused] == [((48, 48, 128), (48, 48, 128), 1, 12), ((24, 24, 256), (24, 24, 256), 1, 12)]
    assert len(auto.layers) == len(base.layers) - 2 and len(every.layers) == len(base.layers) - 11
    assert sum(L.kind == lowering.OP_DWCONV3X3 for L in every.layers) == 0
    assert auto.flops_per_image() == base.flops_per_image() == every.flops_per_image()
    assert base.bytes_per_image() - auto.bytes_per_image() == 2 * 4 * (48 * 48 * 128 + 24 * 24 * 256)      # the two depthwise results
    assert "conv_dw_3_relu/clip_by_value" not in auto.tensor_layer and "conv_pw_3_relu/clip_by_value" in auto.tensor_layer
    # a requested depthwise tensor stays materialised; fp32-only arithmetic has nothing to fuse with
    keep = lowering.lower_graph(graph, "input_1:0", {OUT_FEATURES: "global_pooling/Mean:0", OUT_AGE: "conv_dw_3_relu/clip_by_value:0"}, (192, 192))
    assert sum(L.kind == lowering.OP_DWPW_F16S for L in keep.layers) == 1
    assert not any(L.kind == lowering.OP_DWPW_F16S for L in lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192), pw_math="f32").layers)
    with pytest.raises(ValueError):
        lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (64, 64), block_fusion="some")
    import plan_ref
    z = np.load(os.path.join(GOLDEN, "e2e_synthetic.npz"))
    x = np.random.RandomState(123).uniform(-128, 128, (3, 96, 96, 3)).astype(np.float32)
    for mode in ("auto", "all"):
        plan = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96), block_fusion=mode)
        assert lowering.OP_DWPW_F16S in [op[0] for op in plan_ref.parse(plan.serialize())["ops"]]
        assert rel(plan_ref.run(plan.serialize(), x)["features"], z["feat_96"]) < 1e-6


def test_features_only_plan_and_intermediate_outputs(graph):
    plan = lowering.lower_graph(graph, "input_1:0", {OUT_FEATURES: "global_pooling/Mean:0"}, (64, 64))
    assert len(plan.layers) == 23 and OUT_AGE not in plan.outputs      # stem2, pw 2, 11 blocks (two of them one fused layer each), pool
    x = np.random.RandomState(5).uniform(-128, 128, (1, 64, 64, 3)).astype(np.float32)
    ref = tfo.GraphOracle(MODEL_PB).run("global_pooling/Mean:0", {"input_1:0": x})
    assert rel(plan_ref.run(plan.serialize(), x)["features"], ref) < 1e-6
    # an intermediate trunk tensor can be the output too
    plan2 = lowering.lower_graph(graph, "input_1:0", {OUT_FEATURES: "conv_pw_3_relu/clip_by_value:0"}, (64, 64))
    ref2 = tfo.GraphOracle(MODEL_PB).run("conv_pw_3_relu/clip_by_value:0", {"input_1:0": x})
    assert rel(plan_ref.run(plan2.serialize(), x)["features"], ref2.reshape(1, -1)) < 1e-6
    # a pre-activation tensor alone is fine (the layer simply ends there) ...
    plan3 = lowering.lower_graph(graph, "input_1:0", {OUT_FEATURES: "conv_pw_3_bn/batchnorm_1/add_1:0"}, (64, 64))
    assert plan3.layers[-1].act == lowering.ACT_NONE
    ref3 = tfo.GraphOracle(MODEL_PB).run("conv_pw_3_bn/batchnorm_1/add_1:0", {"input_1:0": x})
    assert rel(plan_ref.run(plan3.serialize(), x)["features"], ref3.reshape(1, -1)) < 1e-6
    # ... but not together with a later tensor of the same fused layer
    with pytest.raises(lowering.LoweringError):
        lowering.lower_graph(graph, "input_1:0", {OUT_FEATURES: "conv_pw_3_bn/batchnorm_1/add_1:0",
                                                  OUT_AGE: "conv_pw_3_relu/clip_by_value:0"}, (64, 64))


def test_plan_struct_layout_matches_header():
    hdr = open(os.path.join(os.path.dirname(MODEL_PB), "..", "include", "hsefr.h")).read()
    assert "HSEFR_PLAN_MAGIC 0x314c505246455348ull" in hdr
    assert lowering._HEADER.size == 64 and lowering._BUFFER.size == 16 and lowering._OP.size == 112
    assert struct.pack("<Q", lowering.PLAN_MAGIC) == b"HSEFRPL1"
    for name, val in (("HSEFR_OP_CONV_C3", lowering.OP_CONV_C3), ("HSEFR_OP_DWCONV3X3", lowering.OP_DWCONV3X3),
                      ("HSEFR_OP_PWCONV_F32", lowering.OP_PWCONV_F32), ("HSEFR_OP_GAP", lowering.OP_GAP),
                      ("HSEFR_OP_DENSE", lowering.OP_DENSE), ("HSEFR_OP_SOFTMAX", lowering.OP_SOFTMAX),
                      ("HSEFR_OP_DWPW_F32", lowering.OP_DWPW_F32), ("HSEFR_OP_PWCONV_F16S", lowering.OP_PWCONV_F16S),
                      ("HSEFR_OP_DWPW_F16S", lowering.OP_DWPW_F16S), ("HSEFR_OP_STEM_F16S", lowering.OP_STEM_F16S),
                      ("HSEFR_OP_STEM2_F16S", lowering.OP_STEM2_F16S),
                      ("HSEFR_ACT_RELU6", lowering.ACT_RELU6), ("HSEFR_ACT_SIGMOID", lowering.ACT_SIGMOID)):
        assert "%s = %d" % (name, val) in hdr


def test_buffers_never_alias_input_and_output(graph):
    plan = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192))
    for L in plan.layers:
        if L.src >= 0:
            assert plan.layers[L.src].out_buf != L.out_buf
    pinned = {plan.layers[li].out_buf for li, _ in plan.outputs.values()}
    assert len(pinned) == 3
    # the two big ping-pong buffers + three small output buffers
    assert sorted(plan.buffers, reverse=True)[:2] == [4 * 48 * 48 * 128, 4 * 48 * 48 * 128]      # nothing at 96x96 is materialised any more
    unfused = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192), fuse=False)
    assert sorted(unfused.buffers, reverse=True)[:2] == [4 * 96 * 96 * 64, 4 * 96 * 96 * 32]


def test_unfolded_batchnorm_and_learning_phase_graph():
    """vgg2_mobilenet.pb (missing) keeps BN un-folded behind a keras_learning_phase Switch/Merge
    (facerec_test.py:212).  Build a tiny graph of that shape by hand and lower it."""
    import gb
    rs = np.random.RandomState(11)
    b = gb.GraphBuilder()
    b.placeholder("input_1", [-1, 8, 8, 3])
    b.placeholder("conv1_bn/keras_learning_phase", None, dtype=10)
    k = (rs.randn(3, 3, 3, 8) * 0.1).astype(np.float32)
    b.const("conv1/kernel", k)
    b.node("conv1/convolution", "Conv2D", ["input_1", "conv1/kernel"], strides=[1, 2, 2, 1], padding="SAME", data_format="NHWC")
    gamma, beta = rs.uniform(0.5, 1.5, 8).astype(np.float32), rs.randn(8).astype(np.float32)
    mean, var = rs.randn(8).astype(np.float32), rs.uniform(0.5, 2, 8).astype(np.float32)
    for nm, v in (("gamma", gamma), ("beta", beta), ("moving_mean", mean), ("moving_variance", var)):
        b.const("conv1_bn/" + nm, v)
    b.const("conv1_bn/eps", np.float32(1e-3))
    # inference branch: (x - mean) * gamma * rsqrt(var + eps) + beta, as Keras emits it
    b.node("conv1_bn/cond/Switch_1", "Switch", ["conv1/convolution", "conv1_bn/keras_learning_phase"])
    b.node("conv1_bn/batchnorm/add", "Add", ["conv1_bn/moving_variance", "conv1_bn/eps"])
    b.node("conv1_bn/batchnorm/Rsqrt", "Rsqrt", ["conv1_bn/batchnorm/add"])
    b.node("conv1_bn/batchnorm/mul", "Mul", ["conv1_bn/batchnorm/Rsqrt", "conv1_bn/gamma"])
    b.node("conv1_bn/batchnorm/mul_1", "Mul", ["conv1_bn/cond/Switch_1", "conv1_bn/batchnorm/mul"])
    b.node("conv1_bn/batchnorm/mul_2", "Mul", ["conv1_bn/moving_mean", "conv1_bn/batchnorm/mul"])
    b.node("conv1_bn/batchnorm/sub", "Sub", ["conv1_bn/beta", "conv1_bn/batchnorm/mul_2"])
    b.node("conv1_bn/batchnorm/add_1", "Add", ["conv1_bn/batchnorm/mul_1", "conv1_bn/batchnorm/sub"])
    # training branch hangs off port 1 and must be dead
    b.node("conv1_bn/cond/train", "Neg", ["conv1_bn/cond/Switch_1:1"])
    b.node("conv1_bn/cond/Merge", "Merge", ["conv1_bn/batchnorm/add_1", "conv1_bn/cond/train"])
    b.node("conv1_relu/Relu6", "Relu6", ["conv1_bn/cond/Merge"])
    b.const("gap/axes", np.array([1, 2], np.int32))
    b.node("global_average_pooling2d_1/Mean", "Mean", ["conv1_relu/Relu6", "gap/axes"])
    b.const("reshape_1/shape", np.array([-1, 1, 1, 8], np.int32))
    b.node("reshape_1/Reshape", "Reshape", ["global_average_pooling2d_1/Mean", "reshape_1/shape"])
    data = b.serialize()

    g = graphdef.read_graph(data)
    plan = lowering.lower_graph(g, "input_1:0", {OUT_FEATURES: "reshape_1/Reshape:0"}, None,
                                {"conv1_bn/keras_learning_phase:0": 0})
    assert [L.kind for L in plan.layers] == [lowering.OP_CONV_C3, lowering.OP_GAP]
    assert plan.layers[0].act == lowering.ACT_RELU6
    x = rs.uniform(-128, 128, (2, 8, 8, 3)).astype(np.float32)
    got = plan_ref.run(plan.serialize(), x)["features"]
    # oracle: same bytes through the unfused interpreter, learning phase fed 0 (facerec_test.py:118-119)
    ref = tfo.GraphOracle(tfo.parse_graphdef(data)).run("reshape_1/Reshape:0",
                                                        {"input_1:0": x, "conv1_bn/keras_learning_phase:0": 0})
    assert ref.shape == (2, 1, 1, 8)
    assert rel(got, ref.reshape(2, -1)) < 1e-5
    with pytest.raises(lowering.LoweringError):        # predicate not fed -> cannot resolve the Merge
        lowering.lower_graph(g, "input_1:0", {OUT_FEATURES: "reshape_1/Reshape:0"})


def test_pointwise_math_selection_and_split_weight_image(graph):
    """auto: every pointwise layer of the MobileNet trunk reads a ReLU6 output -> split-f16 products (wire kind 12);
    'f32' keeps the fp32 MFMA kernel; the split image reproduces the fp32 kernel to 2^-21 and inverts exactly."""
    import plan_ref
    plan = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96), block_fusion="none")
    pws = [L for L in plan.layers if L.kind == lowering.OP_PWCONV_F32]
    assert len(pws) == 12 and all(L.a_log2 == 12 for L in pws)          # pw_2 .. pw_13 (pw_1 lives in the fused stem)
    wire = [o[0] for o in plan_ref.parse(plan.serialize())["ops"]]
    # pw_2 reads the fused stem's fp32 output, pw_3 / pw_4 are shallow (K = 128): wire kind 12; the nine layers with
    # K >= 256 read a depthwise layer that stores its result pre-split for them (wire kind 16, round 2) -- and that
    # depthwise op carries the split exponent
    assert wire.count(lowering.OP_PWCONV_F16S) == 3 and wire.count(lowering.OP_PWCONV_PS) == 9 and lowering.OP_PWCONV_F32 not in wire
    ops_ = plan_ref.parse(plan.serialize())["ops"]
    assert sum(1 for o in ops_ if o[0] == lowering.OP_DWCONV3X3 and o[16] == 12) == 9
    plain = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96), block_fusion="none", presplit="none")
    wire0 = [o[0] for o in plan_ref.parse(plain.serialize())["ops"]]
    assert wire0.count(lowering.OP_PWCONV_F16S) == 12 and lowering.OP_PWCONV_PS not in wire0
    plan32 = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96), pw_math="f32")
    assert all(L.a_log2 == 0 for L in plan32.layers)
    assert lowering.OP_PWCONV_F16S not in [o[0] for o in plan_ref.parse(plan32.serialize())["ops"]]
    with pytest.raises(ValueError):
        lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96), pw_math="bf16")
    for L in pws:
        w_t = np.ascontiguousarray(L.w.reshape(L.w.shape[2], L.w.shape[3]).T)
        img, descale = lowering.split_pointwise_weights(w_t, 12)
        assert img.dtype == np.uint16 and img.shape == (w_t.shape[0], w_t.shape[1] // 32, 64) and img.nbytes == w_t.nbytes
        e = np.log2(descale.astype(np.float64))
        assert np.array_equal(e, np.round(e))                                   # exact powers of two
        hi = img[:, :, :32].copy().view(np.float16).astype(np.float64)
        assert np.isfinite(hi).all() and np.abs(hi).max() < 16384.5
        amax = np.abs(hi).reshape(w_t.shape[0], -1).max(axis=1)
        assert (amax[np.abs(w_t).max(axis=1) > 0] >= 8191.5).all()              # every channel sits at the top of f16's range
        back = lowering.unsplit_pointwise_weights(img, descale, 12)
        assert np.abs(back - w_t).max() <= 2.0 ** -21 * np.abs(w_t).max(axis=1).max()
        assert (np.abs(back - w_t) <= 2.0 ** -21 * np.abs(w_t).max(axis=1, keepdims=True)).all()
    z = np.zeros((64, 32), np.float32)
    img, descale = lowering.split_pointwise_weights(z)
    assert not img.any() and np.isfinite(descale).all()
    with pytest.raises(lowering.LoweringError):
        lowering.split_pointwise_weights(np.zeros((64, 48), np.float32))


def test_full_keras_mobilenet_graph_of_the_missing_files_shape():
    """The whole 13-block graph in the form of the reference's missing vgg2_mobilenet.pb (facerec_test.py:212): un-folded
    BatchNormalization behind keras_learning_phase Switch/Merge after EVERY convolution, Relu6 ops, reshape_1/Reshape output.
    It must lower to the very plan the shipped (folded, quantised) trunk lowers to, and compute the same function."""
    import keras_mobilenet_graph as kg
    from conftest import MODEL_PB
    data = kg.build(MODEL_PB, 64)
    g = graphdef.read_graph(data)
    feeds = {"conv1_bn/keras_learning_phase:0": 0}
    plan = lowering.lower_graph(g, "input_1:0", {OUT_FEATURES: "reshape_1/Reshape:0"}, None, feeds)
    shipped = lowering.lower_graph(graphdef.read_graph(MODEL_PB), "input_1:0", {OUT_FEATURES: "global_pooling/Mean:0"}, (64, 64))
    assert [L.kind for L in plan.layers] == [L.kind for L in shipped.layers]
    assert [(L.a_log2, L.out_split, L.in_split) for L in plan.layers] == [(L.a_log2, L.out_split, L.in_split) for L in shipped.layers]
    x = np.random.RandomState(0).uniform(-128, 128, (2, 64, 64, 3)).astype(np.float32)
    got = plan_ref.run(plan.serialize(), x)["features"]
    ref = tfo.GraphOracle(tfo.parse_graphdef(data)).run("reshape_1/Reshape:0", {"input_1:0": x, "conv1_bn/keras_learning_phase:0": 0})
    assert ref.shape == (2, 1, 1, 1024)
    assert rel(got, ref.reshape(2, -1)) < 1e-5
    assert rel(got, plan_ref.run(shipped.serialize(), x)["features"]) < 1e-5          # == the shipped trunk's function
    with pytest.raises(lowering.LoweringError):        # learning phase not fed
        lowering.lower_graph(g, "input_1:0", {OUT_FEATURES: "reshape_1/Reshape:0"})


def test_declared_input_bound_selects_the_bounded_stem(graph):
    """lower_graph(input_bound=...) turns the fused stem into wire kind 17 (conv1 on the f16 MFMA, csrc/stem3_fused.hip): the
    conv kernel travels as split rows behind the fp32 pack, the scale exponent is the largest that keeps bound * 2^in_log2
    within f16; the plan computes the same function; without a bound nothing changes."""
    import plan_ref
    plan = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96), input_bound=256.0)
    assert plan.layers[0].kind == lowering.OP_STEM3_F16S and plan.layers[0].in_log2 == 7
    assert lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96), input_bound=151.1).layers[0].in_log2 == 7
    assert lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96), input_bound=1.0).layers[0].in_log2 == 14
    assert lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96), input_bound=1000.0).layers[0].in_log2 == 5
    assert lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96)).layers[0].kind == lowering.OP_STEM2_F16S
    ops_ = plan_ref.parse(plan.serialize())["ops"]
    assert ops_[0][0] == 17 and ops_[0][16] == (12 | ((7 + 64) << 8))
    x = np.random.RandomState(3).randint(0, 256, (2, 96, 96, 3)).astype(np.float32) - np.array([103.939, 116.779, 123.68], np.float32)
    got = plan_ref.run(plan.serialize(), x)
    ref = plan_ref.run(lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96)).serialize(), x)
    for k in ref:
        assert rel(got[k], ref[k]) < 1e-5, k
    with pytest.raises(AssertionError):                    # the CPU plan checker enforces the declared bound too
        plan_ref.run(plan.serialize(), x * 3)
    with pytest.raises(ValueError):
        lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96), input_bound=-1.0)


def test_pointwise_depthwise_epilogue_fusion(graph):
    """fuse_pwdw: on the 12x12 and 6x6 maps (whole images inside a 288-row GEMM tile) the depthwise layer behind a pre-split
    pointwise layer runs in that GEMM's epilogue: six depthwise launches and six fp32 tensors less at 192x192, same graph."""
    import plan_ref
    base = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192), input_bound=256.0, pwdw_fusion="none")
    fused = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192), input_bound=256.0, pwdw_fusion="auto")
    kb, kf = [L.kind for L in base.layers], [L.kind for L in fused.layers]
    assert kf.count(lowering.OP_PWDW_PS) == 7 and kf.count(lowering.OP_PWGAP_PS) == 1 and len(kf) == len(kb) - 8       # dw_7 .. dw_11, dw_12 (stride 2), dw_13; the pool
    assert kb.count(lowering.OP_DWCONV3X3) - kf.count(lowering.OP_DWCONV3X3) == 7
    for L in fused.layers:
        if L.kind == lowering.OP_PWDW_PS:
            assert L.in_split and L.out_split == 12 and L.a_log2 == 12 and 288 % (L.in_shape[0] * L.in_shape[1]) == 0
            assert (L.in_shape[0], L.in_shape[1]) == (L.out_shape[0] * L.stride, L.out_shape[1] * L.stride)
            assert L.out_shape[2] % 128 == 0 and L.w3.shape[:2] == (3, 3)
            assert L.stride == 1 or (L.in_shape[:2] in ((12, 12), (14, 14)) and (L.pad_t, L.pad_l) == (0, 0))
            nxt = [M for M in fused.layers if M.src >= 0 and fused.layers[M.src] is L]
            assert len(nxt) == 1 and nxt[0].in_split            # its split rows feed the next pre-split GEMM
    # the fused layer stands for the DEPTHWISE tensor; the pointwise tensor in between no longer exists
    assert "conv_dw_8_relu/clip_by_value" in fused.tensor_layer and "conv_pw_7_relu/clip_by_value" not in fused.tensor_layer
    assert fused.flops_per_image() == base.flops_per_image()
    assert base.bytes_per_image() - fused.bytes_per_image() == 2 * 4 * (6 * 12 * 12 * 512 + 2 * 6 * 6 * 1024)
    assert fused.layers[fused.tensor_layer["global_pooling/Mean"]].kind == lowering.OP_PWGAP_PS and "conv_pw_13_relu/clip_by_value" not in fused.tensor_layer
    # a requested pointwise tensor keeps its pair unfused
    keep = lowering.lower_graph(graph, "input_1:0", {OUT_FEATURES: "global_pooling/Mean:0", OUT_AGE: "conv_pw_8_relu/clip_by_value:0"}, (192, 192),
                                input_bound=256.0, pwdw_fusion="auto")
    assert [L.kind for L in keep.layers].count(lowering.OP_PWDW_PS) == 6
    # 224-pixel input (the placeholder's own size): 14x14 maps ride in 224-row tiles, 7x7 maps five to a 256-row tile -- the six
    # stride-1 depthwise layers behind pre-split GEMMs, the 14x14 -> 7x7 stride-2 one and the pool fuse
    k224 = [L.kind for L in lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (224, 224), pwdw_fusion="auto").layers]
    assert k224.count(lowering.OP_PWDW_PS) == 7 and k224.count(lowering.OP_PWGAP_PS) == 1
    with pytest.raises(ValueError):
        lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (192, 192), pwdw_fusion="all")
    # the serialised plan evaluates to the same graph (fp64 executor of the wire format)
    x = np.random.RandomState(5).uniform(-120, 130, (1, 96, 96, 3)).astype(np.float32)
    p96 = lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96), input_bound=256.0, pwdw_fusion="auto")
    assert [L.kind for L in p96.layers].count(lowering.OP_PWDW_PS) == 6          # 6x6 maps (five 512-channel blocks) and the 3x3 map (288 % 9 == 0)
    got = plan_ref.run(p96.serialize(), x)
    want = plan_ref.run(lowering.lower_graph(graph, "input_1:0", ALL_OUTS, (96, 96), fuse=False).serialize(), x)
    for k in want:
        assert rel(got[k], want[k]) < 1e-5


def test_plan_describe_routes_the_baseline_plans_to_product_kernels_only():
    """Plan.describe() = layer -> kernel family<template arguments>, answered by libhsefr's own launchers with the
    launch suppressed (no GPU needed).  The four BASELINE plans (MobileNet-192 batch 256, ResNet-50 bf16 batch 128, age / gender 224
    batch 512 with three outputs, and the small-batch plan of the per-image calls) route only to families the product library lists,
    every layer has a kernel or is covered by a flagged layer's launch, and the development-only families are not in the product."""
    import subprocess
    from conftest import MODEL_PB
    from hse_facerec_tf_amd import _lib, graphdef, lowering, resnet50
    g = graphdef.read_graph(MODEL_PB)
    outs = {0: "global_pooling/Mean:0", 1: "age_pred/Softmax:0", 2: "gender_pred/Sigmoid:0"}
    plans = {
        "configs[1]": (lowering.lower_graph(g, "input_1:0", {0: outs[0]}, (192, 192), input_bound=256.0, u8_mean_bgr=(103.939, 116.779, 123.68)), 256),
        "configs[2]": (resnet50.build_plan(resnet50.synthetic_weights(1), (224, 224), "caffe"), 128),
        "configs[3]": (lowering.lower_graph(g, "input_1:0", outs, (224, 224), input_bound=256.0), 512),
        "per-image": (lowering.lower_graph(g, "input_1:0", outs, (224, 224), input_bound=256.0, presplit="none"), 1),
    }
    seen = {}
    for name, (plan, n) in plans.items():
        rows = plan.describe(n)
        assert [r["layer"] for r in rows] == list(range(len(plan.layers)))
        for r in rows:
            assert (r["inside"] is None) == bool(r["kernels"]), (name, r)
            if r["inside"] is not None:
                assert plan.layers[r["inside"]].flags != 0 and r["inside"] < r["layer"]
            for fam in r["family"]:
                assert fam in lowering.PRODUCT_KERNEL_FAMILIES, (name, r)
        seen[name] = rows
    fam = lambda name: [f for r in seen[name] for f in r["family"]]
    assert fam("configs[1]")[0] == "stem5_stream_kernel" and fam("configs[1]").count("pwconv_ps_kernel") == 8
    assert seen["configs[1]"][0]["kernels"] == ["stem5_stream_kernel<2, false>"]
    # (13 of the 16 3x3 layers on the window kernel: the last block of the 56-, 28- and 14-pixel stages runs its 3x3 at stride 2 -- round 6,
    # lowering.subsample_stage_tails -- on the gather kernel)
    assert fam("configs[2]").count("conv1x1_pair_bf16_kernel") == 2 and fam("configs[2]").count("conv3x3_w2_bf16_kernel") == 13
    assert seen["configs[2]"][0]["kernels"] == ["stem7s_stream_kernel"]          # the streaming stem, not the patch kernel it falls back to
    assert fam("configs[3]")[-1] == "heads_kernel" and [r["inside"] for r in seen["configs[3]"][-3:]] == [len(plans["configs[3]"][0].layers) - 4] * 3
    assert "pwconv_ps_kernel" not in fam("per-image") and "heads_kernel" in fam("per-image")
    L = _lib.lib()
    if not hasattr(L, "hsefr_debug_set"):
        syms = subprocess.run(["nm", "-C", _lib.LIB_PATH], capture_output=True, text=True).stdout
        for dev_only in ("stem_fused_kernel", "conv3x3_win_bf16_kernel"):
            assert dev_only not in syms, dev_only
