"""GPU parity for the bf16 ResNet-50 path (BASELINE config 3) vs oracle/resnet50.py.
Tolerance: tensors are STORED as bf16 (8-bit mantissa, ulp = 2^-8 relative), accumulation is fp32 on
the GPU and exact in the oracle, so a stored value may differ by one bf16 ulp where the pre-rounding
sum sits next to a rounding boundary.  Per kernel: |got-want| <= 2^-7*|want| + 2^-9*max|want|; end to
end (53 rounded layers deep): 2e-2 of the feature's max magnitude (~1e-2 for bf16)."""
import numpy as np
import pytest

from oracle import resnet50 as ores
from oracle import tf_graph as tfo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from hse_facerec_tf_amd import ops, resnet50
    assert torch.cuda.is_available()
    return torch, ops, resnet50


def to_dev_bf16(torch, ops, resnet50, a):
    return ops.bf16_from_bits(resnet50.to_bf16_bits(a))


def close_bf16(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    tol = 2.0 ** -7 * np.abs(want) + 2.0 ** -9 * np.abs(want).max()
    bad = np.abs(got - want) > tol
    return not bad.any(), float((np.abs(got - want) / (np.abs(want).max() + 1e-30)).max())


@pytest.mark.parametrize("n,h,w,c,cout,k,s,res,act", [
    (2, 14, 14, 64, 64, 1, 1, False, 1), (2, 14, 14, 64, 256, 1, 1, True, 1), (1, 28, 28, 256, 128, 1, 2, False, 1),
    (2, 13, 11, 128, 128, 3, 1, False, 1), (1, 56, 56, 64, 64, 3, 1, False, 1), (3, 7, 7, 512, 512, 3, 1, False, 1),
    (2, 7, 7, 512, 2048, 1, 1, True, 1), (1, 15, 15, 256, 512, 1, 2, False, 0), (1, 9, 9, 1024, 256, 1, 1, False, 1),
    (5, 5, 5, 64, 192, 3, 1, True, 0),
    # the LDS-DMA implicit GEMM (csrc/conv_dma_bf16.hip): stride-2 projection from 512 channels, many tiles per workgroup with a
    # ragged last tile, residual + ReLU on a 3x3, a 5x5 kernel, stride 2 with padding
    (2, 9, 9, 512, 128, 1, 2, False, 0), (37, 14, 14, 64, 256, 3, 1, True, 1), (3, 11, 13, 128, 64, 5, 1, False, 1),
    (2, 12, 12, 64, 128, 3, 2, False, 1),
    # the window 3x3 kernel (csrc/conv3x3_win_bf16.hip; maps at least 40 wide): its four tile shapes, ragged image groups, residual
    (2, 6, 40, 64, 128, 3, 1, True, 1), (1, 4, 48, 128, 256, 3, 1, False, 1), (3, 2, 44, 64, 64, 3, 1, False, 0),
    (1, 28, 56, 64, 64, 3, 1, True, 1), (2, 5, 41, 192, 192, 3, 1, False, 1),
    # the four-wave window 3x3 kernel (csrc/conv3x3_w2_bf16.hip, round 5): its three geometries (rows of <= 16 / 32 / 64 pixels),
    # several channel slabs, residual, heights that are not a multiple of the tile's rows, columns dropped at the right edge,
    # more tiles than workgroups (a persistent workgroup walks two tiles)
    (2, 14, 14, 256, 256, 3, 1, False, 1), (3, 13, 12, 64, 128, 3, 1, True, 1), (1, 15, 16, 128, 128, 3, 1, False, 0),
    (2, 28, 28, 128, 128, 3, 1, False, 1), (1, 9, 25, 64, 128, 3, 1, True, 1), (1, 30, 32, 64, 256, 3, 1, False, 1),
    (2, 6, 50, 128, 64, 3, 1, True, 1), (1, 7, 64, 64, 192, 3, 1, False, 1), (260, 14, 14, 64, 128, 3, 1, False, 1),
    # the four-wave 1x1 GEMM (csrc/conv1x1_w4_bf16.hip, round 5: K-deep reductions with >= 20 000 output pixels): stride 1 and 2 (gathered
    # rows), a ragged last tile, several tiles per workgroup
    (103, 14, 14, 256, 128, 1, 1, False, 1), (30, 53, 54, 256, 128, 1, 2, False, 1), (27, 28, 28, 512, 256, 1, 1, False, 0),
    # ... and its FLAT geometry for maps of at most 7 x 7: whole images per tile, ragged image groups, 6-pixel edges, residual
    (6, 7, 7, 128, 128, 3, 1, False, 1), (5, 6, 7, 64, 64, 3, 1, True, 1), (2, 7, 6, 128, 192, 3, 1, False, 0), (131, 7, 7, 64, 128, 3, 1, True, 1)])
def test_conv_bf16_vs_oracle(env, n, h, w, c, cout, k, s, res, act):
    torch, ops, resnet50 = env
    rs = np.random.RandomState(h * 7 + c + cout + k)
    x = ores.bf16_round(rs.uniform(0, 2, (n, h, w, c)))
    kern = (rs.randn(k, k, c, cout) * np.sqrt(2.0 / (k * k * c))).astype(np.float32)
    sc = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rs.randn(cout).astype(np.float32) * 0.1
    pad = (k - 1) // 2
    y = tfo.conv2d(x, ores.bf16_round(kern), (s, s), "", explicit_pads=(pad,) * 4) * sc + sh
    y = ores.bf16_round(y)
    r = None
    if res:
        r = ores.bf16_round(rs.uniform(-1, 1, y.shape))
        y = y + r
    if act == 1:
        y = np.maximum(y, 0)
    want = ores.bf16_round(y)
    d = lambda a: to_dev_bf16(torch, ops, resnet50, a.astype(np.float32))
    got = ops.conv_bf16(d(x), ops.bf16_from_bits(resnet50.pack_conv_weight(kern)), torch.from_numpy(sc).cuda(),
                        torch.from_numpy(sh).cuda(), k, k, s, pad, None if r is None else d(r), act)
    assert tuple(got.shape) == want.shape
    ok, err = close_bf16(got.float().cpu().numpy(), want)
    assert ok, "max rel err %.3e" % err


@pytest.mark.parametrize("n,hw,c,cout,res,act", [(128, 56, 64, 256, False, 0), (128, 56, 64, 64, False, 1), (128, 56, 64, 256, True, 1),
                                                 (128, 56, 256, 64, False, 1), (128, 28, 128, 512, True, 1), (128, 14, 1024, 256, False, 1),
                                                 (128, 7, 512, 2048, True, 1), (3, 9, 64, 192, True, 1), (1, 5, 128, 64, False, 0),
                                                 # (the four-wave GEMM with RESIDENT weight stages: K = 256 / 128, one and two channel tiles)
                                                 (128, 28, 256, 128, False, 1), (128, 28, 256, 256, False, 0), (96, 28, 256, 128, False, 1)])
def test_conv1x1_persistent_kernel_bit_for_bit_and_run_to_run(env, n, hw, c, cout, res, act):
    """The persistent 1x1 kernel (conv1x1_bf16.hip) must reproduce the general implicit-GEMM kernel BIT FOR BIT on every
    1x1 stride-1 layer shape of ResNet-50 at batch 128, three launches in a row (an unguarded store-data hazard -- DESIGN.md
    lesson 14 -- once gave rare wrong dwords at exactly these sizes, K = 64 without residual, and nowhere smaller).
    The general kernel is reached through the same entry point with the SAME convolution written as a 3x3 / pad 1 kernel
    whose eight outer taps are zero: identical channel blocking, and adding exact zeros changes no fp32 sum."""
    torch, ops, resnet50 = env
    g = torch.Generator(device="cuda").manual_seed(hw * 31 + c + cout)
    x = torch.randn((n, hw, hw, c), device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn((cout, c), device="cuda", generator=g) / c ** 0.5).to(torch.bfloat16)
    sc = torch.rand((cout,), device="cuda", generator=g) + 0.5
    sh = torch.randn((cout,), device="cuda", generator=g)
    r = torch.randn((n, hw, hw, cout), device="cuda", generator=g).to(torch.bfloat16) if res else None
    w3 = torch.zeros((cout, 3, 3, c), device="cuda", dtype=torch.bfloat16)
    w3[:, 1, 1, :] = w
    ref = ops.conv_bf16(x, w3.reshape(cout, 9 * c).contiguous(), sc, sh, 3, 3, 1, 1, r, act)
    for _ in range(3):
        assert torch.equal(ops.conv_bf16(x, w, sc, sh, 1, 1, 1, 0, r, act), ref)


@pytest.mark.parametrize("n,oh,ow,c,cout,c2,s2,h2,w2,act", [
    (2, 14, 14, 64, 256, 64, 1, 14, 14, 1), (1, 14, 14, 128, 512, 256, 2, 28, 28, 1), (3, 7, 7, 256, 1024, 512, 2, 14, 14, 1),
    (2, 4, 4, 512, 2048, 1024, 2, 7, 7, 1), (5, 9, 11, 64, 192, 128, 2, 17, 22, 0), (37, 13, 13, 64, 128, 64, 1, 13, 13, 1),
    # (>= 192 tiles of 224 x 128 and K + K2 >= 256: csrc/conv1x1_w4_bf16.hip PROJ -- the stage-entry blocks of the 28-, 14- and 7-pixel
    # stages with a ragged last tile each, an odd stride-2 view, a stride-1 projection; and one shape just under the tile bound)
    (14, 28, 28, 128, 512, 256, 2, 56, 56, 1), (28, 14, 14, 256, 1024, 512, 2, 28, 28, 1), (56, 7, 7, 512, 2048, 1024, 2, 14, 14, 1),
    (30, 27, 27, 128, 256, 128, 2, 53, 54, 0), (110, 14, 14, 192, 256, 64, 1, 14, 14, 1), (22, 14, 14, 128, 512, 256, 2, 28, 28, 1)])
def test_increase_layer_with_projected_shortcut_vs_oracle_and_vs_the_two_launch_form(env, n, oh, ow, c, cout, c2, s2, h2, w2, act):
    """csrc/conv1x1_bf16.hip PROJ (round 5): act(bf16(s W x + b) + bf16(s2 W2 x2[::s2] + b2)) in one launch -- against the oracle's
    two convolutions with the projection's tensor rounded to bf16 in between, and BIT FOR BIT against the two launches it replaces
    (same K order per product, same rounding points -- also where the fused pair runs on another kernel family than the two launches).  Shapes: the four stage-entry blocks of ResNet-50 in small,
    an odd-sized stride-2 view (h2 = 2 oh - 1), a ragged last tile, several tiles per workgroup."""
    torch, ops, resnet50 = env
    rs = np.random.RandomState(oh * 7 + c + cout + c2)
    x = ores.bf16_round(rs.uniform(0, 2, (n, oh, ow, c)))
    x2 = ores.bf16_round(rs.uniform(0, 2, (n, h2, w2, c2)))
    k1 = (rs.randn(1, 1, c, cout) * np.sqrt(2.0 / c)).astype(np.float32)
    k2 = (rs.randn(1, 1, c2, cout) * np.sqrt(2.0 / c2)).astype(np.float32)
    sc, sh = rs.uniform(0.5, 1.5, cout).astype(np.float32), (rs.randn(cout) * 0.1).astype(np.float32)
    sc2, sh2 = rs.uniform(0.5, 1.5, cout).astype(np.float32), (rs.randn(cout) * 0.1).astype(np.float32)
    proj = ores.bf16_round(tfo.conv2d(x2, ores.bf16_round(k2), (s2, s2), "", explicit_pads=(0,) * 4) * sc2 + sh2)
    assert proj.shape == (n, oh, ow, cout)
    ya = ores.bf16_round(tfo.conv2d(x, ores.bf16_round(k1), (1, 1), "", explicit_pads=(0,) * 4) * sc + sh)
    y = ya + proj
    want = ores.bf16_round(np.maximum(y, 0) if act == 1 else y)
    d = lambda a: to_dev_bf16(torch, ops, resnet50, a.astype(np.float32))
    f = lambda a: torch.from_numpy(a).cuda()
    pk = lambda k: ops.bf16_from_bits(resnet50.pack_conv_weight(k))
    got = ops.conv1x1_proj_bf16(d(x), pk(k1), f(sc), f(sh), d(x2), pk(k2), f(sc2), f(sh2), s2, act)
    # each ADDEND is a stored bf16 value and may sit one ulp from the oracle's where its fp32 sum is next to a rounding boundary;
    # where the two nearly cancel that ulp is large against the result: close_bf16's bound + one ulp of each addend
    g64 = got.float().cpu().numpy().astype(np.float64)
    tol = 2.0 ** -7 * np.abs(want) + 2.0 ** -9 * np.abs(want).max() + 2.0 ** -8 * (np.abs(ya) + np.abs(proj))
    assert (np.abs(g64 - want) <= tol).all(), "max rel err %.3e" % (np.abs(g64 - want) / np.abs(want).max()).max()
    p_dev = ops.conv_bf16(d(x2), pk(k2), f(sc2), f(sh2), 1, 1, s2, 0, None, 0)
    two = ops.conv_bf16(d(x), pk(k1), f(sc), f(sh), 1, 1, 1, 0, p_dev, act)
    # (the shapes with >= 192 tiles of 224 x 128 and K + K2 >= 256 run on csrc/conv1x1_w4_bf16.hip's PROJ kernel -- 16x16x32 MFMAs where the
    # two-launch form's kernels use 32x32x16 -- and are still bit-identical: the matrix pipe adds a K-step's products in K order either way)
    assert torch.equal(got, two)
    assert torch.equal(ops.conv1x1_proj_bf16(d(x), pk(k1), f(sc), f(sh), d(x2), pk(k2), f(sc2), f(sh2), s2, act), got)     # run to run


@pytest.mark.parametrize("n,hw,c,cout,c2", [(128, 56, 64, 256, 64), (100, 56, 64, 128, 64), (70, 28, 64, 256, 64)])
def test_register_staged_kernels_with_resident_weight_tiles_bit_for_bit(env, n, hw, c, cout, c2):
    """the register-staged 1x1 kernel keeps its weight tile RESIDENT in LDS when a tile's K loop is at most two steps and the
    grid's stride keeps a workgroup on one channel origin (more tiles than slots).  Nothing small reaches that state: here the plain kernel
    with a residual (K = 64) and the PROJ kernel (K + K2 = 128) at shapes with more than 512 tiles of 128 x 128, both outputs bit for bit
    against the general implicit-GEMM kernel (the same convolution written as a 3 x 3 whose outer taps are zero) resp. the two-launch form."""
    torch, ops, resnet50 = env
    g = torch.Generator(device="cuda").manual_seed(hw * 31 + n + cout)
    x = torch.randn((n, hw, hw, c), device="cuda", generator=g).to(torch.bfloat16)
    x2 = torch.randn((n, hw, hw, c2), device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn((cout, c), device="cuda", generator=g) / c ** 0.5).to(torch.bfloat16)
    w2 = (torch.randn((cout, c2), device="cuda", generator=g) / c2 ** 0.5).to(torch.bfloat16)
    sc, sh = torch.rand((cout,), device="cuda", generator=g) + 0.5, torch.randn((cout,), device="cuda", generator=g)
    sc2, sh2 = torch.rand((cout,), device="cuda", generator=g) + 0.5, torch.randn((cout,), device="cuda", generator=g)
    assert ((n * hw * hw + 127) // 128) * (cout // 128) > 512
    r = torch.randn((n, hw, hw, cout), device="cuda", generator=g).to(torch.bfloat16)
    w3 = torch.zeros((cout, 3, 3, c), device="cuda", dtype=torch.bfloat16)
    w3[:, 1, 1, :] = w
    ref = ops.conv_bf16(x, w3.reshape(cout, 9 * c).contiguous(), sc, sh, 3, 3, 1, 1, r, 1)
    for _ in range(2):
        assert torch.equal(ops.conv_bf16(x, w, sc, sh, 1, 1, 1, 0, r, 1), ref)
    pr = ops.conv_bf16(x2, w2, sc2, sh2, 1, 1, 1, 0, None, 0)
    two = ops.conv_bf16(x, w, sc, sh, 1, 1, 1, 0, pr, 1)
    for _ in range(2):
        assert torch.equal(ops.conv1x1_proj_bf16(x, w, sc, sh, x2, w2, sc2, sh2, 1, 1), two)


def test_conv_bf16_exact_integers(env):
    """Small integers are exact in bf16 and fp32: any im2col / fragment-map / swizzle mix-up shows as an exact mismatch."""
    torch, ops, resnet50 = env
    rs = np.random.RandomState(1)
    for (n, h, w, c, cout, k, s) in [(1, 6, 5, 64, 64, 3, 1), (2, 8, 8, 128, 128, 1, 2), (1, 4, 4, 64, 128, 3, 1),
                                     (1, 14, 14, 128, 128, 3, 1), (1, 28, 28, 64, 128, 3, 1), (1, 8, 56, 64, 64, 3, 1), (3, 7, 7, 64, 64, 3, 1)]:    # (conv3x3_w2)
        x = rs.randint(-2, 3, (n, h, w, c)).astype(np.float64)
        kern = rs.randint(-1, 2, (k, k, c, cout)).astype(np.float32)
        pad = (k - 1) // 2
        want = tfo.conv2d(x, kern.astype(np.float64), (s, s), "", explicit_pads=(pad,) * 4)
        assert np.abs(want).max() < 256                      # exactly representable in bf16
        d = lambda a: to_dev_bf16(torch, ops, resnet50, a.astype(np.float32))
        got = ops.conv_bf16(d(x), ops.bf16_from_bits(resnet50.pack_conv_weight(kern)), torch.ones(cout, device="cuda"),
                            torch.zeros(cout, device="cuda"), k, k, s, pad, None, 0)
        assert np.array_equal(got.float().cpu().numpy().astype(np.float64), want)


@pytest.mark.parametrize("n,hw", [(2, 64), (1, 224), (3, 37)])
def test_stem_vs_oracle(env, n, hw):
    torch, ops, resnet50 = env
    rs = np.random.RandomState(hw)
    x = rs.uniform(-130, 150, (n, hw, hw, 3)).astype(np.float32)
    kern = (rs.randn(7, 7, 3, 64) * 0.02).astype(np.float32)
    sc = rs.uniform(0.5, 1.5, 64).astype(np.float32)
    sh = rs.randn(64).astype(np.float32)
    want = ores.bf16_round(np.maximum(tfo.conv2d(ores.bf16_round(x), ores.bf16_round(kern), (2, 2), "", explicit_pads=(3, 3, 3, 3)) * sc + sh, 0))
    got = ops.stem7x7_bf16(torch.from_numpy(x).cuda(), ops.bf16_from_bits(resnet50.pack_stem_weight(kern)),
                           torch.from_numpy(sc).cuda(), torch.from_numpy(sh).cuda())
    assert tuple(got.shape) == want.shape
    ok, err = close_bf16(got.float().cpu().numpy(), want)
    assert ok, "max rel err %.3e" % err


@pytest.mark.parametrize("h,w,ceil", [(112, 112, True), (112, 112, False), (9, 7, True), (8, 8, True)])
def test_maxpool_bit_exact(env, h, w, ceil):
    torch, ops, resnet50 = env
    x = ores.bf16_round(np.random.RandomState(h).randn(2, h, w, 64))
    want = ores._maxpool_3x3_s2(x, "caffe" if ceil else "valid")
    got = ops.maxpool3x3s2_bf16(to_dev_bf16(torch, ops, resnet50, x.astype(np.float32)), ceil)
    assert np.array_equal(got.float().cpu().numpy().astype(np.float64), want)


def test_gap_bf16(env):
    torch, ops, resnet50 = env
    x = ores.bf16_round(np.random.RandomState(2).uniform(0, 4, (3, 7, 7, 2048)))
    got = ops.gap_bf16(to_dev_bf16(torch, ops, resnet50, x.astype(np.float32))).cpu().numpy()
    assert np.abs(got - x.mean(axis=(1, 2))).max() < 2e-6 * np.abs(x).max()


@pytest.mark.parametrize("size,pool,n", [(64, "caffe", 2), (96, "valid", 1), (224, "caffe", 1)])
def test_resnet50_end_to_end_vs_oracle(env, size, pool, n):
    torch, ops, resnet50 = env
    w = resnet50.synthetic_weights(123)
    ext = resnet50.ResNet50Extractor(w, (size, size), max_batch=4, pool=pool)
    x = np.random.RandomState(size).uniform(-120, 130, (n, size, size, 3)).astype(np.float32)
    got = ext.extract_batch(torch.from_numpy(x).cuda()).cpu().numpy()
    want = ores.forward(w, x, pool)
    assert got.shape == want.shape == (n, 2048)
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err < 2e-2, err
    # direction of the embedding (what 1-NN uses) is preserved far better than the bar
    cos = (got * want).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(want, axis=1)
    assert cos.min() > 0.9995
    ext.close_session()


def test_resnet50_batch_128_properties(env):
    """BASELINE config 3 at full size (batch 128, 224x224x3): batch-position independence, bit-exact."""
    torch, ops, resnet50 = env
    ext = resnet50.ResNet50Extractor(None, (224, 224), max_batch=128)
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.uniform(-120, 130, (128, 224, 224, 3)).astype(np.float32)).cuda()
    full = ext.extract_batch(x)
    assert tuple(full.shape) == (128, 2048) and bool(torch.isfinite(full).all()) and float(full.min()) >= 0
    perm = torch.from_numpy(rs.permutation(128)).cuda()
    assert torch.equal(ext.extract_batch(x[perm].contiguous()), full[perm])
    assert torch.equal(ext.extract_batch(x[40:43].contiguous()), full[40:43])
    ext.close_session()


@pytest.mark.parametrize("pool,bn,head,hw", [("SAME", "fused", "avgpool", 40), ("PADVALID", "muladd", "mean", 38)])
def test_resnet_style_frozen_graph_through_the_dropin_class(env, tmp_path, pool, bn, head, hw):
    """A ResNet-style .pb (shape of the reference's missing vgg2_resnet.pb) through TensorFlowInference with the
    registry arguments of facerec_test.py:213; dtype is picked automatically (bf16 MFMA path)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    import mini_resnet_graph
    from hse_facerec_tf_amd import TensorFlowInference
    torch, ops, resnet50 = env
    data, dim = mini_resnet_graph.build(3, hw, pool, bn, 64, head)
    pb = tmp_path / "mini_resnet.pb"
    pb.write_bytes(data)
    tfi = TensorFlowInference(str(pb), input_tensor='input:0', output_tensor='pool5_7x7_s1:0', convert2BGR=True,
                              imageNetUtilsMean=False, max_batch=4)
    assert tfi.dtype == "bf16" and (tfi.w, tfi.h) == (hw, hw) and tfi.feature_dim == dim
    x = np.random.RandomState(1).uniform(-100, 120, (3, hw, hw, 3)).astype(np.float32)
    got = tfi.extract_batch(x)
    want = tfo.GraphOracle(tfo.parse_graphdef(data), np.float64).run("pool5_7x7_s1:0", {"input:0": x}).reshape(3, -1)
    assert got.shape == want.shape
    assert np.abs(got - want).max() / np.abs(want).max() < 2e-2
    tfi.close_session()


@pytest.mark.parametrize("n,hw,c,cout,k,stride,pad,res,act", [(2, 9, 8, 12, 3, 1, 1, True, 1), (1, 23, 3, 64, 7, 2, 3, False, 1),
                                                                (3, 7, 64, 256, 1, 1, 0, True, 0), (2, 14, 20, 8, 1, 2, 0, False, 2)])
def test_conv2d_f32_vs_oracle(env, n, hw, c, cout, k, stride, pad, res, act):
    """The general exact-fp32 convolution (OP_CONV_F32) against the fp64 oracle: 1e-6 (fp32 accumulation of k*k*c terms)."""
    torch, ops, resnet50 = env
    rs = np.random.RandomState(hw * 7 + c)
    x = rs.uniform(-3, 3, (n, hw, hw, c)).astype(np.float32)
    kern = (rs.randn(k, k, c, cout) / np.sqrt(k * k * c)).astype(np.float32)
    sc, sh = rs.uniform(0.5, 1.5, cout).astype(np.float32), rs.randn(cout).astype(np.float32)
    want = tfo.conv2d(x.astype(np.float64), kern.astype(np.float64), (stride, stride), "", explicit_pads=(pad, pad, pad, pad)) * sc + sh
    r = rs.randn(*want.shape).astype(np.float32) if res else None
    if res:
        want = want + r
    want = np.maximum(want, 0) if act else want
    want = np.minimum(want, 6) if act == 2 else want
    got = ops.conv2d_f32(torch.from_numpy(x).cuda(), torch.from_numpy(kern).cuda(), torch.from_numpy(sc).cuda(), torch.from_numpy(sh).cuda(),
                         stride=stride, pad=pad, res=None if r is None else torch.from_numpy(r).cuda(), act=act).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-6 * max(np.abs(want).max(), 1.0)
    with pytest.raises(Exception):
        ops.conv2d_f32(torch.from_numpy(x).cuda(), torch.from_numpy(kern[..., :cout - 1].copy()).cuda())      # cout % 4 != 0


@pytest.mark.parametrize("n,h,w,c,cout,k,stride,pad,res,act", [
    (2, 9, 11, 16, 64, 3, 1, 1, True, 1),        # 3x3, one K run per tap, BN = 64, ragged M (198 pixels = 1.5 tiles)
    (1, 23, 23, 3, 64, 7, 2, 3, False, 1),       # the 7x7/2 stem: value-by-value gather, K = 147 -> 10 steps with a zero tail
    (3, 7, 7, 64, 256, 1, 1, 0, True, 1),        # 1x1 "increase" + residual + ReLU, BN = 128, two N tiles
    (2, 14, 14, 256, 128, 1, 2, 0, False, 0),    # stride-2 1x1 projection, linear
    (1, 10, 6, 128, 128, 3, 2, 1, False, 2),     # 3x3 stride 2 (odd output), ReLU6
    (5, 8, 8, 32, 192, 3, 1, 1, True, 0)])       # cout = 192 -> BN = 64 x 3
def test_conv2d_f32_on_the_fp32_matrix_pipe(env, n, h, w, c, cout, k, stride, pad, res, act):
    """csrc/conv_f32_mfma.hip (what OP_CONV_F32 runs for cout % 64 == 0) against the fp64 oracle at the direct kernel's bar, and
    against the direct vector-FMA kernel itself; three launches bit-identical."""
    torch, ops, resnet50 = env
    rs = np.random.RandomState(h * 7 + c + cout)
    x = rs.uniform(-3, 3, (n, h, w, c)).astype(np.float32)
    kern = (rs.randn(k, k, c, cout) / np.sqrt(k * k * c)).astype(np.float32)
    sc, sh = rs.uniform(0.5, 1.5, cout).astype(np.float32), rs.randn(cout).astype(np.float32)
    want = tfo.conv2d(x.astype(np.float64), kern.astype(np.float64), (stride, stride), "", explicit_pads=(pad, pad, pad, pad)) * sc + sh
    r = rs.randn(*want.shape).astype(np.float32) if res else None
    if res:
        want = want + r
    want = np.maximum(want, 0) if act else want
    want = np.minimum(want, 6) if act == 2 else want
    args = (torch.from_numpy(x).cuda(), torch.from_numpy(kern).cuda(), torch.from_numpy(sc).cuda(), torch.from_numpy(sh).cuda())
    kw = dict(stride=stride, pad=pad, res=None if r is None else torch.from_numpy(r).cuda(), act=act)
    got = [ops.conv2d_f32(*args, mfma=True, **kw) for _ in range(3)]
    assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2])
    g = got[0].cpu().numpy()
    assert g.shape == want.shape
    assert np.abs(g - want).max() < 2e-6 * max(np.abs(want).max(), 1.0)
    direct = ops.conv2d_f32(*args, **kw).cpu().numpy()
    assert np.abs(g - direct).max() < 2e-6 * max(np.abs(want).max(), 1.0)
    with pytest.raises(NotImplementedError):
        ops.conv2d_f32(args[0], torch.from_numpy(kern[..., :cout - 4].copy()).cuda(), mfma=True)      # cout % 64 != 0


@pytest.mark.parametrize("size,pool,n", [(64, "caffe", 2), (224, "caffe", 1)])
def test_resnet50_fp32_grade_mode_meets_the_1e4_bar(env, size, pool, n):
    """ResNet-50 in the fp32-grade mode against the exact (unrounded) oracle at the bar BASELINE states
    for features (max abs err <= 1e-4 relative to the feature scale); the bf16 mode on the same input for scale."""
    torch, ops, resnet50 = env
    w = resnet50.synthetic_weights(123)
    x = np.random.RandomState(size).uniform(-120, 130, (n, size, size, 3)).astype(np.float32)
    want = ores.forward(w, x, pool, storage="exact")
    ext = resnet50.ResNet50Extractor(w, (size, size), max_batch=4, pool=pool, dtype="f32")
    got = ext.extract_batch(torch.from_numpy(x).cuda()).cpu().numpy()
    ext.close_session()
    assert got.shape == want.shape == (n, 2048)
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err < 1e-5, err          # the synthetic weights give features of magnitude ~1e3: the 1e-4 bar scaled by max|feature|, with 10x margin
    ext16 = resnet50.ResNet50Extractor(w, (size, size), max_batch=4, pool=pool)
    e16 = np.abs(ext16.extract_batch(torch.from_numpy(x).cuda()).cpu().numpy() - want).max() / np.abs(want).max()
    ext16.close_session()
    assert err < e16 / 50


@pytest.mark.parametrize("pool,bn,head,hw", [("SAME", "fused", "avgpool", 40), ("PADVALID", "muladd", "mean", 38)])
def test_resnet_style_frozen_graph_in_the_fp32_grade_mode(env, tmp_path, pool, bn, head, hw):
    """TensorFlowInference(dtype='f32') on a ResNet-style .pb: the MobileNet kernels do not cover it, so the general
    exact-fp32 lowering runs (tfi.dtype == 'f32g'); 1e-5 against the fp64 graph."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    import mini_resnet_graph
    from hse_facerec_tf_amd import TensorFlowInference
    torch, ops, resnet50 = env
    data, dim = mini_resnet_graph.build(3, hw, pool, bn, 64, head)
    pb = tmp_path / "mini_resnet.pb"
    pb.write_bytes(data)
    tfi = TensorFlowInference(str(pb), input_tensor='input:0', output_tensor='pool5_7x7_s1:0', convert2BGR=True,
                              imageNetUtilsMean=False, max_batch=4, dtype="f32")
    assert tfi.dtype == "f32g" and tfi.feature_dim == dim
    x = np.random.RandomState(1).uniform(-100, 120, (3, hw, hw, 3)).astype(np.float32)
    got = tfi.extract_batch(x)
    want = tfo.GraphOracle(tfo.parse_graphdef(data), np.float64).run("pool5_7x7_s1:0", {"input:0": x}).reshape(3, -1)
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-5
    tfi.close_session()


@pytest.mark.parametrize("n,h,w,ceil,ppad", [(2, 64, 64, True, 0), (1, 224, 224, True, 0), (1, 224, 224, False, 0), (3, 37, 51, True, 0),
                                              (2, 38, 38, False, 1), (1, 70, 45, False, 0), (2, 30, 29, True, 1), (2, 7, 9, True, 0)])
def test_fused_stem_pool_vs_oracle(env, n, h, w, ceil, ppad):
    """conv1 + ReLU + pool1 in one kernel (csrc/stem7s_stream.hip; the 7 x 9 image is below its minimum and runs the patch kernel,
    csrc/stem7x7_pool.hip) against the oracle's conv -> bf16 -> clipped max-pool, and
    against the two-kernel path (same rounding points; fp32 accumulation order differs: rare one-ulp bf16 differences)."""
    torch, ops, resnet50 = env
    rs = np.random.RandomState(h * 3 + w)
    x = rs.uniform(-130, 150, (n, h, w, 3)).astype(np.float32)
    kern = (rs.randn(7, 7, 3, 64) * 0.02).astype(np.float32)
    sc = rs.uniform(0.5, 1.5, 64).astype(np.float32)
    sh = rs.randn(64).astype(np.float32)
    c1 = ores.bf16_round(np.maximum(tfo.conv2d(ores.bf16_round(x), ores.bf16_round(kern), (2, 2), "", explicit_pads=(3, 3, 3, 3)) * sc + sh, 0))
    oh, ow = c1.shape[1:3]
    ph = (-(-(oh + 2 * ppad - 3) // 2) if ceil else (oh + 2 * ppad - 3) // 2) + 1
    pw = (-(-(ow + 2 * ppad - 3) // 2) if ceil else (ow + 2 * ppad - 3) // 2) + 1
    pb, pr = max((ph - 1) * 2 + 3 - oh - ppad, 0), max((pw - 1) * 2 + 3 - ow - ppad, 0)
    xp = np.pad(c1, ((0, 0), (ppad, pb), (ppad, pr), (0, 0)), constant_values=-np.inf)
    want = np.full((n, ph, pw, 64), -np.inf)
    for dy in range(3):
        for dx in range(3):
            want = np.maximum(want, xp[:, dy:dy + 2 * (ph - 1) + 1:2, dx:dx + 2 * (pw - 1) + 1:2, :])
    xd, wd = torch.from_numpy(x).cuda(), ops.bf16_from_bits(resnet50.pack_stem_weight(kern))
    scd, shd = torch.from_numpy(sc).cuda(), torch.from_numpy(sh).cuda()
    got = ops.stem7x7_pool_bf16(xd, wd, scd, shd, ceil_mode=ceil, pool_pad=ppad)
    assert tuple(got.shape) == want.shape
    ok, err = close_bf16(got.float().cpu().numpy(), want)
    assert ok, "max rel err %.3e" % err
    if ppad == 0:
        two = ops.maxpool3x3s2_bf16(ops.stem7x7_bf16(xd, wd, scd, shd), ceil)
        d = (got.float() - two.float()).abs()
        assert float((d > 0).float().mean()) < 2e-3 and bool((d <= 0.0079 * two.float().abs() + 1e-6).all())      # <= one bf16 ulp, rarely


@pytest.mark.parametrize("n,oh,ow,c,cout,st,h2,w2", [(2, 28, 28, 64, 256, 2, 56, 56), (3, 14, 14, 128, 512, 2, 28, 28), (5, 7, 7, 256, 1024, 2, 14, 14),
                                                       (2, 13, 9, 64, 128, 2, 25, 17), (1, 5, 6, 64, 64, 3, 13, 16), (37, 14, 14, 64, 256, 2, 28, 27),
                                                       (128, 28, 28, 64, 256, 2, 56, 56), (17, 128, 128, 64, 64, 2, 256, 255)])
def test_conv1x1_with_a_strided_residual_equals_the_gathered_form(env, n, oh, ow, c, cout, st, h2, w2):
    """hsefr_conv1x1_sres_bf16 (the increase layer of a stage's last block on the compact map, lowering.subsample_stage_tails): the residual
    read at every st-th pixel of a larger map -- BIT FOR BIT the plain layer on the gathered residual (same kernel, same K order; only the
    residual's addresses differ), ragged last tiles and odd maps included; the last case has rows x pixels-per-map beyond 2^32 (the row ->
    (image, y, x) quotients must be exact for every 32-bit row, csrc/common.h hsefr_udiv)."""
    torch, ops, resnet50 = env
    g = torch.Generator(device="cuda").manual_seed(n * 131 + oh)
    x = (torch.rand((n, oh, ow, c), device="cuda", generator=g) * 2 - 0.5).to(torch.bfloat16)
    r = (torch.randn((n, h2, w2, cout), device="cuda", generator=g)).to(torch.bfloat16)
    w = (torch.randn((cout, c), device="cuda", generator=g) / c ** 0.5).to(torch.bfloat16)
    sc = torch.rand(cout, device="cuda", generator=g) + 0.5
    sh = torch.randn(cout, device="cuda", generator=g)
    rg = r[:, ::st, ::st, :][:, :oh, :ow, :].contiguous()
    for act in (1, 0):
        got = ops.conv1x1_sres_bf16(x, w, sc, sh, r, st, act)
        want = ops.conv_bf16(x, w, sc, sh, 1, 1, res=rg, act=act)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    with pytest.raises(Exception):
        ops.conv1x1_sres_bf16(x, w, sc, sh, r[:, :(oh - 1) * st, :, :].contiguous(), st)      # the map does not reach the last output row


@pytest.mark.parametrize("n,h,w,proj", [(2, 56, 56, False), (3, 28, 20, True), (1, 13, 9, False), (128, 56, 56, False), (5, 7, 31, True), (3, 256, 250, False)])
def test_pair_with_its_first_output_stored_at_even_pixels(env, n, h, w, proj):
    """hsefr_conv1x1_pair_sub2_bf16 (HSEFR_OPF_OUT_SUB2): y1 stored at even rows / columns only -- bit for bit y1[:, ::2, ::2] of the plain pair,
    y2 (computed from every pixel, in registers) unchanged; odd maps, ragged tiles, both shortcut forms."""
    torch, ops, resnet50 = env
    g = torch.Generator(device="cuda").manual_seed(h * 7 + w)
    bf = lambda *sh: (torch.randn(sh, device="cuda", generator=g)).to(torch.bfloat16)
    x, w1, w2 = bf(n, h, w, 64), (bf(256, 64).float() / 8).to(torch.bfloat16), (bf(64, 256).float() / 16).to(torch.bfloat16)
    s1, b1, s2, b2 = (torch.rand(256, device="cuda", generator=g) + 0.5, torch.randn(256, device="cuda", generator=g),
                      torch.rand(64, device="cuda", generator=g) + 0.5, torch.randn(64, device="cuda", generator=g))
    kw = dict(x2=bf(n, h, w, 64), wp_packed=(bf(256, 64).float() / 8).to(torch.bfloat16), scale_p=s1.clone(), shift_p=b1.clone()) if proj else dict(res=bf(n, h, w, 256))
    y1, y2 = ops.conv1x1_pair_bf16(x, w1, s1, b1, w2, s2, b2, **kw)
    c1, c2 = ops.conv1x1_pair_bf16(x, w1, s1, b1, w2, s2, b2, y1_sub2=True, **kw)
    assert tuple(c1.shape) == (n, (h + 1) // 2, (w + 1) // 2, 256)
    assert torch.equal(c1.view(torch.int16), y1[:, ::2, ::2, :].contiguous().view(torch.int16)) and torch.equal(c2.view(torch.int16), y2.view(torch.int16))


def test_resnet50_with_and_without_subsampled_stage_tails(env):
    """The whole network with the last block of the 56-, 28- and 14-pixel stages computed only where the next stage reads it
    (lowering.subsample_stage_tails, the default) against the plan that computes every pixel: the same features to bf16-pipeline accuracy
    (the stride-2 3x3 layers run on another kernel: another fp32 accumulation order), both within the oracle's tolerance elsewhere."""
    torch, ops, resnet50 = env
    from hse_facerec_tf_amd.engine import Engine
    w = resnet50.synthetic_weights(11)
    x = torch.from_numpy(np.random.RandomState(2).uniform(-120, 130, (6, 224, 224, 3)).astype(np.float32)).cuda()
    outs = []
    for sub in (True, False):
        plan = resnet50.build_plan(w, (224, 224), "caffe", subsample=sub)
        assert sum(L.graph_hw is not None for L in plan.layers) == (7 if sub else 0)      # three 3x3 + three increase layers + the pair output stored compact
        eng = Engine(plan, max_batch=6)
        outs.append(list(eng.forward(x, (0,)).values())[0].float().cpu().numpy())
        eng.close()
    a, b = outs
    assert np.isfinite(a).all() and np.abs(a - b).max() <= 0.02 * np.abs(b).max() and np.abs(a - b).mean() <= 2e-3 * np.abs(b).mean()


def test_fused_stem_pool_many_units_per_wave(env):
    """More strip units than resident waves (300 images x 3 strips x 3 segments = 2700 > 2048): a wave of the streaming stem walks several
    units -- fresh carried rows, fresh window ring -- and the sweep direction flips between launches.  Reference: the two-kernel path
    (conv1 map + max-pool), same rounding points (rare one-ulp bf16 differences: another fp32 accumulation order)."""
    torch, ops, resnet50 = env
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.uniform(-130, 150, (300, 64, 64, 3)).astype(np.float32)).cuda()
    wd = ops.bf16_from_bits(resnet50.pack_stem_weight((rs.randn(7, 7, 3, 64) * 0.02).astype(np.float32)))
    sc, sh = torch.from_numpy(rs.uniform(0.5, 1.5, 64).astype(np.float32)).cuda(), torch.from_numpy(rs.randn(64).astype(np.float32)).cuda()
    two = ops.maxpool3x3s2_bf16(ops.stem7x7_bf16(x, wd, sc, sh), True).float()
    for _ in range(2):
        got = ops.stem7x7_pool_bf16(x, wd, sc, sh).float()
        d = (got - two).abs()
        assert float((d > 0).float().mean()) < 2e-3 and bool((d <= 0.0079 * two.abs() + 1e-6).all())


def test_resnet50_fused_stem_equals_unfused_network(env):
    """The whole network with and without the fused stem: same features to bf16-pipeline accuracy (the only difference is the
    accumulation order inside conv1), and the fused plan is what ResNet50Extractor runs by default."""
    torch, ops, resnet50 = env
    from hse_facerec_tf_amd import lowering
    w = resnet50.synthetic_weights(123)
    x = torch.from_numpy(np.random.RandomState(9).uniform(-120, 130, (3, 224, 224, 3)).astype(np.float32)).cuda()
    a = resnet50.ResNet50Extractor(w, (224, 224), max_batch=4)
    b = resnet50.ResNet50Extractor(w, (224, 224), max_batch=4, fuse=False)
    assert a.plan.layers[0].kind == lowering.OP_STEM7X7_POOL_BF16 and b.plan.layers[0].kind == lowering.OP_STEM7X7_BF16
    fa, fb = a.extract_batch(x), b.extract_batch(x)
    assert float((fa - fb).abs().max() / fb.abs().max()) < 5e-3
    a.close_session(), b.close_session()


@pytest.mark.parametrize("n,h,w,proj,act1,act2", [
    (2, 14, 14, False, 1, 1), (1, 9, 7, True, 1, 1), (3, 5, 5, False, 0, 1), (37, 13, 13, True, 1, 0),
    # more wave tiles than the grid's waves (a wave walks several tiles), a ragged last tile, and the BASELINE shape itself
    (300, 31, 29, False, 1, 1), (128, 56, 56, True, 1, 1), (128, 56, 56, False, 1, 1)])
def test_increase_reduce_pair_vs_oracle_and_bit_for_bit_vs_the_two_launches(env, n, h, w, proj, act1, act2):
    """csrc/conv1x1_pair_bf16.hip (round 6): a 64 -> 256 increase layer (+ residual | + projected shortcut) and the 256 -> 64 reduce layer
    behind it in one launch, the 256-channel tensor chained through registers.  Against the oracle's convolutions with every stored
    tensor rounded to bf16 (small shapes), and BIT FOR BIT, both outputs, against the launches it replaces (every shape)."""
    torch, ops, resnet50 = env
    c, c1, c2o = 64, 256, 64
    rs = np.random.RandomState(h * 7 + w + n)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    pk = lambda k: ops.bf16_from_bits(resnet50.pack_conv_weight(k))
    k1 = (rs.randn(1, 1, c, c1) * np.sqrt(2.0 / c)).astype(np.float32)
    k2 = (rs.randn(1, 1, c1, c2o) * np.sqrt(2.0 / c1)).astype(np.float32)
    kp = (rs.randn(1, 1, 64, c1) * np.sqrt(2.0 / 64)).astype(np.float32)
    sc1, sh1 = rs.uniform(0.5, 1.5, c1).astype(np.float32), (rs.randn(c1) * 0.1).astype(np.float32)
    sc2, sh2 = rs.uniform(0.5, 1.5, c2o).astype(np.float32), (rs.randn(c2o) * 0.1).astype(np.float32)
    scp, shp = rs.uniform(0.5, 1.5, c1).astype(np.float32), (rs.randn(c1) * 0.1).astype(np.float32)
    g = torch.Generator(device="cuda").manual_seed(h * 31 + w)
    x = (torch.rand((n, h, w, c), device="cuda", generator=g) * 2).to(torch.bfloat16)
    x2 = (torch.rand((n, h, w, 64), device="cuda", generator=g) * 2).to(torch.bfloat16)
    r = (torch.rand((n, h, w, c1), device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    if proj:
        y1, y2 = ops.conv1x1_pair_bf16(x, pk(k1), f(sc1), f(sh1), pk(k2), f(sc2), f(sh2), x2=x2, wp_packed=pk(kp), scale_p=f(scp), shift_p=f(shp),
                                       act1=act1, act2=act2)
        ref1 = ops.conv1x1_proj_bf16(x, pk(k1), f(sc1), f(sh1), x2, pk(kp), f(scp), f(shp), 1, act1)
    else:
        y1, y2 = ops.conv1x1_pair_bf16(x, pk(k1), f(sc1), f(sh1), pk(k2), f(sc2), f(sh2), res=r, act1=act1, act2=act2)
        ref1 = ops.conv_bf16(x, pk(k1), f(sc1), f(sh1), 1, 1, 1, 0, r, act1)
    ref2 = ops.conv_bf16(ref1, pk(k2), f(sc2), f(sh2), 1, 1, 1, 0, None, act2)
    assert torch.equal(y1, ref1), "increase output differs from the single launch in %d elements" % int((y1 != ref1).sum())
    assert torch.equal(y2, ref2), "reduce output differs from the single launch in %d elements" % int((y2 != ref2).sum())
    for _ in range(2):      # run to run
        if proj:
            z1, z2 = ops.conv1x1_pair_bf16(x, pk(k1), f(sc1), f(sh1), pk(k2), f(sc2), f(sh2), x2=x2, wp_packed=pk(kp), scale_p=f(scp),
                                           shift_p=f(shp), act1=act1, act2=act2)
        else:
            z1, z2 = ops.conv1x1_pair_bf16(x, pk(k1), f(sc1), f(sh1), pk(k2), f(sc2), f(sh2), res=r, act1=act1, act2=act2)
        assert torch.equal(z1, y1) and torch.equal(z2, y2)
    if n * h * w > 20000:
        return
    xo = x.float().cpu().numpy().astype(np.float64)
    conv = lambda a, k: tfo.conv2d(a, ores.bf16_round(k), (1, 1), "", explicit_pads=(0,) * 4)
    ya = ores.bf16_round(conv(xo, k1) * sc1 + sh1)
    rr = ores.bf16_round(conv(x2.float().cpu().numpy().astype(np.float64), kp) * scp + shp) if proj else r.float().cpu().numpy().astype(np.float64)
    w1 = ya + rr
    want1 = ores.bf16_round(np.maximum(w1, 0) if act1 == 1 else w1)
    g1 = y1.float().cpu().numpy().astype(np.float64)
    tol1 = 2.0 ** -7 * np.abs(want1) + 2.0 ** -9 * np.abs(want1).max() + 2.0 ** -8 * (np.abs(ya) + np.abs(rr))
    assert (np.abs(g1 - want1) <= tol1).all()
    # the second product against the oracle FED WITH THE DEVICE'S y1 (one-ulp differences of y1 would otherwise be compared twice)
    w2 = conv(g1, k2) * sc2 + sh2
    want2 = ores.bf16_round(np.maximum(ores.bf16_round(w2), 0) if act2 == 1 else ores.bf16_round(w2))
    ok, err = close_bf16(y2.float().cpu().numpy(), want2)
    assert ok, "max rel err %.3e" % err


def test_resnet50_plan_pairs_run_as_one_launch_and_change_no_bit(env):
    """lowering.mark_pairs on the ResNet-50 plan: conv2_1 / conv2_2's increase layers carry HSEFR_OPF_PAIR_NEXT, the engine runs each
    with the reduce layer behind it as one launch (the covered op's profiled interval is empty), and the features equal the plan without
    pairs bit for bit (both tensors of a pair against the launches they replace: the kernel test above)."""
    torch, ops, resnet50 = env
    from hse_facerec_tf_amd import lowering
    from hse_facerec_tf_amd.engine import Engine
    w = resnet50.synthetic_weights(5)
    paired = resnet50.build_plan(w, (224, 224), "caffe")
    plain = resnet50.build_plan(w, (224, 224), "caffe", pair=False)
    flagged = [L.name for L in paired.layers if L.flags & lowering.OPF_PAIR_NEXT]
    assert flagged == ["conv2_1_1x1_increase", "conv2_2_1x1_increase"] and not any(L.flags for L in plain.layers)
    x = torch.from_numpy(np.random.RandomState(2).uniform(-120, 130, (6, 224, 224, 3)).astype(np.float32)).cuda()
    ea, eb = Engine(paired, max_batch=6), Engine(plain, max_batch=6)
    fa, fb = ea.forward(x)["features"], eb.forward(x)["features"]
    assert torch.equal(fa, fb)
    ea.set_profiling(1)
    ea.forward(x)
    t = ea.op_times_ms(0)
    for i, L in enumerate(paired.layers):
        if L.flags & lowering.OPF_PAIR_NEXT:
            assert t[i] > t[i + 1] and t[i + 1] < 0.015, (L.name, t[i], t[i + 1])     # (two event records back to back: a few microseconds)
    ea.close()
    eb.close()
