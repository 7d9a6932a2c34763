"""CPU suite, part 6: the ResNet-50 plan builder (topology, weight packing, buffer plan) executed
from its SERIALISED bytes by tests/plan_ref.py and compared with the independently written oracle
(oracle/resnet50.py).  Both emulate bf16 storage, so agreement is expected to ~1e-6."""
import numpy as np
import pytest

import plan_ref
from hse_facerec_tf_amd import resnet50
from oracle import resnet50 as ores


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


def test_topology_counts_and_sizes():
    convs = resnet50.conv_names()
    assert len(convs) == 1 + 3 * 16 + 4                      # stem + 16 bottlenecks x 3 + 4 projections = 53 convs
    params = sum(k * k * ci * co for _, k, ci, co, _ in convs)
    assert abs(params / 1e6 - 23.45) < 0.1                   # 23.45 M parameters (95 MB fp32 file)
    from hse_facerec_tf_amd import lowering
    fused = resnet50.build_plan(resnet50.synthetic_weights(1), (224, 224), "caffe")
    assert fused.layers[0].kind == lowering.OP_STEM7X7_POOL_BF16 and fused.layers[0].out_shape == (56, 56, 64) and fused.layers[0].pad3 == (0, 0)
    assert len(fused.layers) == 50 and fused.layers[1].src == 0      # 55 layers - conv1 (inside the pool) - the four projections
    # round 5: every stage's projected shortcut runs inside its block's increase layer (lowering.fuse_proj), which then reads the block input
    pj = {L.name: L for L in fused.layers if L.proj is not None}
    assert sorted(pj) == ["conv%d_1_1x1_increase" % k for k in (2, 3, 4, 5)] and not any("proj" in L.name for L in fused.layers)
    # round 6: the last block of a stage runs at the pixels the next stage's stride-2 layers read (lowering.subsample_stage_tails), so those
    # layers -- and the projections -- read a compact map at stride 1
    assert pj["conv2_1_1x1_increase"].proj == (64, 1, 56, 56) and pj["conv3_1_1x1_increase"].proj == (256, 1, 28, 28)
    assert pj["conv5_1_1x1_increase"].proj == (1024, 1, 7, 7) and fused.layers[pj["conv3_1_1x1_increase"].res].name == "conv2_3_1x1_increase"
    full = resnet50.build_plan(resnet50.synthetic_weights(1), (224, 224), "caffe", subsample=False)
    pf = {L.name: L for L in full.layers}
    assert pf["conv3_1_1x1_increase"].proj == (256, 2, 56, 56) and pf["conv5_1_1x1_increase"].proj == (1024, 2, 14, 14)
    assert resnet50.flops_per_image(full) == resnet50.flops_per_image(fused)      # the algorithmic figures are the GRAPH's
    assert resnet50.activation_bytes_per_image(full) == resnet50.activation_bytes_per_image(fused)
    for L in fused.layers:
        for s_ in (L.src, L.res):
            if s_ >= 0:
                assert fused.layers[s_].out_buf != L.out_buf
    plan = resnet50.build_plan(resnet50.synthetic_weights(1), (224, 224), "caffe", fuse=False)
    assert resnet50.flops_per_image(fused) == resnet50.flops_per_image(plan)
    assert resnet50.activation_bytes_per_image(plan) - resnet50.activation_bytes_per_image(fused) == 2 * 112 * 112 * 64 * 2   # conv1's map: one write, one read
    assert plan.layers[1].out_shape == (56, 56, 64)           # Caffe ceil-mode pool: 112 -> 56
    assert plan.layers[-2].out_shape == (7, 7, 2048) and plan.layers[-1].out_shape == (1, 1, 2048)
    assert abs(resnet50.flops_per_image(plan) / 1e9 - 7.71) < 0.05     # 7.71 GFLOP / image
    assert abs(resnet50.activation_bytes_per_image(plan) / 1e6 - 54.7) < 8   # ~54.7 MB / image in bf16
    valid = resnet50.build_plan(resnet50.synthetic_weights(1), (224, 224), "valid")
    assert valid.layers[0].out_shape == (55, 55, 64)          # keras_vggface valid pool: 112 -> 55
    # stride placement: on the first 1x1 (reduce) and on the projection of stages 3-5
    by = {L.name: L for L in plan.layers}
    assert by["conv3_1_1x1_reduce"].stride == 2 and by["conv3_1_1x1_proj"].stride == 2 and by["conv3_1_3x3"].stride == 1
    assert by["conv2_1_1x1_reduce"].stride == 1 and by["conv3_2_1x1_reduce"].stride == 1
    for L in plan.layers:                                      # no op writes a buffer it reads
        for s in (L.src, L.res):
            if s >= 0:
                assert plan.layers[s].out_buf != L.out_buf


def test_bf16_bit_conversion_matches_oracle_rounding():
    rs = np.random.RandomState(0)
    a = np.concatenate([rs.randn(1000).astype(np.float32) * 10, np.float32([0, 1, -1, 1.00390625, 1.01171875, 3.0e38])])
    bits = resnet50.to_bf16_bits(a)
    back = (bits.astype(np.uint32) << 16).view(np.float32)
    assert np.array_equal(back.astype(np.float64), ores.bf16_round(a))
    assert back[-3] == np.float32(1.0) and back[-2] == np.float32(1.015625)     # ties to even


@pytest.mark.parametrize("size,pool,fuse", [(64, "caffe", True), (70, "valid", True), (64, "caffe", False), (70, "valid", False)])
def test_plan_equals_oracle(size, pool, fuse):
    w = resnet50.synthetic_weights(7)
    x = np.random.RandomState(3).uniform(-120, 130, (1, size, size, 3)).astype(np.float32)
    plan = resnet50.build_plan(w, (size, size), pool, fuse=fuse)
    assert len(plan.layers) == (50 if fuse else 55)          # fuse: conv1 inside the pool, the four projections inside their increase layers
    got = plan_ref.run(plan.serialize(), x)["features"]
    want = ores.forward(w, x, pool)
    assert got.shape == want.shape == (1, 2048)
    assert np.isfinite(want).all() and np.abs(want).max() > 1e-3
    assert rel(got, want) < 1e-6


@pytest.mark.parametrize("pool,bn,head,hw", [("SAME", "fused", "avgpool", 40), ("PADVALID", "muladd", "mean", 38), ("SAME", "fused", "avgpool", 38)])
def test_generic_lowering_of_a_resnet_style_graph(pool, bn, head, hw):
    """a Caffe-converted ResNet-style frozen graph lowers to the bf16 plan generically (Pad+VALID
    stem, FusedBatchNorm or Mul/Add, residual Add fused into the conv epilogue, max-pool, global pool)."""
    import mini_resnet_graph
    from hse_facerec_tf_amd import graphdef, lowering
    from oracle import tf_graph as tfo
    data, dim = mini_resnet_graph.build(3, hw, pool, bn, 64, head)
    g = graphdef.read_graph(data)
    fused = lowering.lower_graph(g, "input:0", {0: "pool5_7x7_s1:0"}, dtype="bf16")
    x = np.random.RandomState(1).uniform(-100, 120, (2, hw, hw, 3)).astype(np.float32)
    plan = lowering.lower_graph(g, "input:0", {0: "pool5_7x7_s1:0"}, dtype="bf16", fuse=False)
    assert fused.layers[0].kind == lowering.OP_STEM7X7_POOL_BF16 and len(fused.layers) == len(plan.layers) - 1 - 2      # conv1, two projections
    assert sorted(L.name for L in fused.layers if L.proj is not None) == ["conv2_1_1x1_increase", "conv3_1_1x1_increase"]
    fl = {L.name: L for L in fused.layers}
    # the stride-2 projection of the second stage reads the first stage's last block, which runs at the pixels it reads (round 6)
    assert fl["conv3_1_1x1_increase"].proj[1] == 1 and fl["conv3_1_1x1_reduce"].stride == 1 and fl["conv2_2_3x3"].stride == 2
    # ... and its shortcut -- the output of the pair in front of it -- is stored at the pixels it reads (lowering.compact_pair_outputs)
    assert fl["conv2_1_1x1_increase"].flags == lowering.OPF_PAIR_NEXT | lowering.OPF_OUT_SUB2 and fl["conv2_1_1x1_increase"].out_shape == fl["conv2_2_1x1_increase"].out_shape
    assert fl["conv2_2_1x1_increase"].res_geom is None and fl["conv2_2_1x1_reduce"].in_shape[:2] == fl["conv2_1_1x1_increase"].graph_hw
    nolaunch = lowering.lower_graph(g, "input:0", {0: "pool5_7x7_s1:0"}, dtype="bf16", launch_fusion=False)
    nl = {L.name: L for L in nolaunch.layers}
    assert nl["conv2_2_1x1_increase"].res_geom == (2,) + nl["conv2_1_1x1_increase"].out_shape[:2] and nl["conv2_1_1x1_increase"].flags == 0
    assert np.array_equal(plan_ref.run(nolaunch.serialize(), x)["features"], plan_ref.run(fused.serialize(), x)["features"])
    assert fused.layers[0].pad3 == (plan.layers[1].pad_t, plan.layers[1].pad_l) == ((1, 1) if (pool, hw) == ("SAME", 38) else (0, 0))
    assert np.array_equal(plan_ref.run(fused.serialize(), x)["features"], plan_ref.run(plan.serialize(), x)["features"])
    # conv1's tensor requested as an output keeps the stem unfused
    both = lowering.lower_graph(g, "input:0", {0: "pool5_7x7_s1:0", 1: "conv1/relu:0"}, dtype="bf16")
    assert both.layers[0].kind == lowering.OP_STEM7X7_BF16
    kinds = [L.kind for L in plan.layers]
    assert kinds[0] == lowering.OP_STEM7X7_BF16 and kinds[1] == lowering.OP_MAXPOOL_BF16 and kinds[-1] == lowering.OP_GAP_BF16
    assert kinds.count(lowering.OP_CONV_BF16) == 3 * 3 + 2          # 3 bottlenecks x 3 convs + 2 projections
    by = {L.name: L for L in plan.layers}
    assert by["conv2_1_1x1_increase"].res == [i for i, L in enumerate(plan.layers) if L.name == "conv2_1_1x1_proj"][0]
    assert by["conv2_2_1x1_increase"].res >= 0 and by["conv2_2_1x1_increase"].act == lowering.ACT_RELU
    assert by["conv3_1_1x1_reduce"].stride == 2 and by["conv1/7x7_s2"].pad_t == 3
    x = np.random.RandomState(1).uniform(-100, 120, (2, hw, hw, 3)).astype(np.float32)
    got = plan_ref.run(plan.serialize(), x)["features"]
    want = tfo.GraphOracle(tfo.parse_graphdef(data), np.float64).run("pool5_7x7_s1:0", {"input:0": x}).reshape(2, -1)
    assert got.shape == want.shape == (2, dim)
    assert rel(got, want) < 2e-2                                     # bf16 storage (plan) vs exact fp64 graph (oracle)
    with pytest.raises(lowering.LoweringError):                      # the fp32 MobileNet kernels do not cover this graph
        lowering.lower_graph(g, "input:0", {0: "pool5_7x7_s1:0"})


@pytest.mark.parametrize("size,pool", [(64, "caffe"), (70, "valid")])
def test_fp32_grade_plan_equals_exact_oracle(size, pool):
    """the fp32-grade mode of the ResNet builder -- same topology on the exact-fp32 general kernels,
    checked against the oracle WITHOUT bf16 storage emulation."""
    from hse_facerec_tf_amd import lowering
    w = resnet50.synthetic_weights(7)
    x = np.random.RandomState(3).uniform(-120, 130, (1, size, size, 3)).astype(np.float32)
    plan = resnet50.build_plan(w, (size, size), pool, dtype="f32")
    kinds = [L.kind for L in plan.layers]
    assert kinds[0] == lowering.OP_CONV_F32 and kinds[1] == lowering.OP_MAXPOOL_F32 and kinds[-1] == lowering.OP_GAP
    assert kinds.count(lowering.OP_CONV_F32) == 53 and all(L.out_bytes == 4 * int(np.prod(L.out_shape)) for L in plan.layers)
    assert resnet50.flops_per_image(plan) == resnet50.flops_per_image(resnet50.build_plan(w, (size, size), pool))
    got = plan_ref.run(plan.serialize(), x)["features"]
    want = ores.forward(w, x, pool, storage="exact")
    assert rel(got, want) < 1e-12
    assert 1e-4 < rel(ores.forward(w, x, pool), want) < 2e-2      # what the bf16 pipeline gives up, for scale
    with pytest.raises(ValueError):
        resnet50.build_plan(w, (size, size), pool, dtype="f16")


@pytest.mark.parametrize("pool,bn,head,hw", [("SAME", "fused", "avgpool", 40), ("PADVALID", "muladd", "mean", 38)])
def test_generic_lowering_of_a_resnet_style_graph_to_the_fp32_grade_plan(pool, bn, head, hw):
    import mini_resnet_graph
    from hse_facerec_tf_amd import graphdef, lowering
    from oracle import tf_graph as tfo
    data, dim = mini_resnet_graph.build(3, hw, pool, bn, 64, head)
    g = graphdef.read_graph(data)
    plan = lowering.lower_graph(g, "input:0", {0: "pool5_7x7_s1:0"}, dtype="f32g")
    kinds = [L.kind for L in plan.layers]
    assert kinds[0] == lowering.OP_CONV_F32 and kinds[1] == lowering.OP_MAXPOOL_F32 and kinds[-1] == lowering.OP_GAP
    assert kinds.count(lowering.OP_CONV_F32) == 1 + 3 * 3 + 2
    by = {L.name: L for L in plan.layers}
    assert by["conv2_1_1x1_increase"].res == [i for i, L in enumerate(plan.layers) if L.name == "conv2_1_1x1_proj"][0]
    x = np.random.RandomState(1).uniform(-100, 120, (2, hw, hw, 3)).astype(np.float32)
    got = plan_ref.run(plan.serialize(), x)["features"]
    want = tfo.GraphOracle(tfo.parse_graphdef(data), np.float64).run("pool5_7x7_s1:0", {"input:0": x}).reshape(2, -1)
    assert got.shape == want.shape == (2, dim)
    assert rel(got, want) < 1e-6          # fp32 constants of the folded BatchNorm vs the fp64 graph
    with pytest.raises(ValueError):
        lowering.lower_graph(g, "input:0", {0: "pool5_7x7_s1:0"}, dtype="fp32")


def test_projected_shortcut_geometry_of_a_256_pixel_block_input_serializes():
    """aux = c2 | stride2 << 12 | h2 << 14 | w2 << 23 reaches bit 31 of the signed wire field from w2 = 256 on (a ResNet-50
    about 1021 px wide): the plan must still serialize, and the reader's masks must give the geometry back."""
    import struct
    from hse_facerec_tf_amd import lowering
    plan = resnet50.build_plan(resnet50.synthetic_weights(1), (1024, 1024), "caffe", subsample=False)
    pj = [L for L in plan.layers if L.proj is not None]
    assert pj[0].proj == (64, 1, 256, 256) and pj[1].proj == (256, 2, 256, 256)

    def words(plan, want_w2):
        data = plan.serialize()
        n_buf, n_ops = struct.unpack_from("<II", data, 12)
        seen = []
        for i in range(n_ops):
            f = lowering._OP.unpack_from(data, lowering._HEADER.size + n_buf * lowering._BUFFER.size + i * lowering._OP.size)
            if f[0] == lowering.OP_CONV_BF16 and (f[-2] != lowering.NO_OFFSET) == want_w2 and f[16] != 0:
                r = f[16]
                seen.append((r & 0xFFF, (r >> 12) & 3, (r >> 14) & 0x1FF, (r >> 23) & 0x1FF))
        return seen
    assert words(plan, True) == [tuple(L.proj) for L in pj]
    # the strided residual's word (round 6: stride << 12 | h2 << 14 | w2 << 23, no channel count) has the same top bit
    sub = resnet50.build_plan(resnet50.synthetic_weights(1), (1024, 1024), "caffe", pair=False)      # (with pairs the 256-pixel stage's shortcut is stored compact)
    sg = [L for L in sub.layers if L.res_geom is not None]
    assert sg[0].res_geom == (2, 256, 256) and words(sub, False) == [(0,) + tuple(L.res_geom) for L in sg]


def test_stage_tails_run_at_the_pixels_the_next_stage_reads():
    """lowering.subsample_stage_tails: conv2_3 / conv3_4 / conv4_6 -- the blocks whose output only stride-2 1x1 layers read -- get a stride-2
    3x3 layer and an increase layer on the compact map whose residual is a stride view of the full-size block input; the consumers read the
    compact tensor at stride 1; a requested tensor is left alone; the plan validates; executed flops drop by 9.7 %."""
    from hse_facerec_tf_amd import _lib, lowering
    w = resnet50.synthetic_weights(1)
    plan = resnet50.build_plan(w, (224, 224), "caffe")
    by = {L.name: (i, L) for i, L in enumerate(plan.layers)}
    for blk, nxt, hw in (("conv2_3", "conv3_1", 56), ("conv3_4", "conv4_1", 28), ("conv4_6", "conv5_1", 14)):
        ix, X = by[blk + "_3x3"]
        ii, I = by[blk + "_1x1_increase"]
        c = I.out_shape[2]
        assert X.stride == 2 and (X.kh, X.pad_t, X.pad_l) == (3, 1, 1) and X.out_shape[:2] == (hw // 2, hw // 2) and X.graph_hw == (hw, hw)
        assert I.src == ix and I.in_shape == X.out_shape and I.out_shape == (hw // 2, hw // 2, c) and I.graph_hw == (hw, hw)
        R = plan.layers[I.res]
        if hw == 56:     # the shortcut comes out of a pair launch: stored at the pixels this block reads (lowering.compact_pair_outputs)
            assert I.res_geom is None and R.flags == lowering.OPF_PAIR_NEXT | lowering.OPF_OUT_SUB2 and R.out_shape == I.out_shape and R.graph_hw == (hw, hw)
            assert plan.layers[I.res + 1].src == I.res and plan.layers[I.res + 1].in_shape == (hw, hw, c)      # ... while the pair's second product reads all of it
        else:            # the shortcut: the full-size map of the block before, read at every second pixel
            assert I.res_geom == (2, hw, hw) and R.out_shape == (hw, hw, c) and R.graph_hw is None
        _, R = by[nxt + "_1x1_reduce"]
        _, P = by[nxt + "_1x1_increase"]
        assert R.src == ii and R.stride == 1 and R.in_shape == I.out_shape and P.res == ii and P.proj[1:] == (1, hw // 2, hw // 2)
    assert sum(L.flags != 0 for L in plan.layers) == 2      # the two stage-2 pairs are still pairs
    assert abs(resnet50.executed_flops_per_image(plan) / resnet50.flops_per_image(plan) - 0.9026) < 1e-3
    blob = plan.serialize()
    L = _lib.lib()
    assert L.hsefr_plan_validate(blob, len(blob)) == 0, _lib.last_error()
    # an odd map: the stride-2 consumers read pixels 0, 2, .., the compact size is ceil(h / 2)
    odd = resnet50.build_plan(w, (200, 200), "valid")
    o = {q.name: q for q in odd.layers}
    h = o["conv2_3_1x1_increase"].graph_hw[0]
    assert h % 2 == 1 and o["conv2_3_1x1_increase"].out_shape[0] == (h + 1) // 2 == o["conv3_1_1x1_reduce"].out_shape[0]
    blob = odd.serialize()
    assert L.hsefr_plan_validate(blob, len(blob)) == 0, _lib.last_error()
    # a REQUESTED tensor keeps its shape: the pass takes the keep list
    layers = [lowering.Layer(**{k: getattr(q, k) for k in q.__dataclass_fields__}) for q in resnet50.build_plan(w, (224, 224), "caffe", subsample=False).layers]
    i3 = [i for i, q in enumerate(layers) if q.name == "conv2_3_1x1_increase"][0]
    assert lowering.subsample_stage_tails(layers, [i3]) == 2 and layers[i3].res_geom is None and layers[i3].out_shape == (56, 56, 256)
