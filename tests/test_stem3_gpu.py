"""GPU parity of the bounded-input fused stem (csrc/stem3_fused.hip, round 2): same graph nodes #30-#55 as stem2_fused.hip
(facerec_test.py:120 / facial_analysis.py:109), conv1's products formed on the f16 MFMA from two-term splits because the
caller DECLARES a bound on the input -- against the exact-fp32-conv1 kernel it stands in for, against the unfused kernels,
and against the fp64 oracle; plus the device-side check of the bound."""
import numpy as np
import pytest

from oracle import tf_graph as tfo

pytestmark = pytest.mark.gpu
TOL = 2e-6


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    from hse_facerec_tf_amd import ops
    return torch, ops


def weights(torch, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    cw = torch.randn((3, 3, 3, 32), device="cuda", generator=g) * 0.02
    cw[..., 5] *= 40.0                                   # output channels decades apart (BN scales up to 88 in the real trunk)
    cw[..., 9] *= 0.01
    csh = torch.randn((32,), device="cuda", generator=g)
    k1 = torch.randn((3, 3, 32), device="cuda", generator=g) / 3
    sc1 = torch.rand((32,), device="cuda", generator=g) + 0.5
    sh1 = torch.randn((32,), device="cuda", generator=g) * 0.3
    kp = (torch.randn((64, 32), device="cuda", generator=g) / 32 ** 0.5).cpu().numpy()
    psh = torch.randn((64,), device="cuda", generator=g)
    k2 = torch.randn((3, 3, 64), device="cuda", generator=g) / 3
    sc2 = torch.rand((64,), device="cuda", generator=g) + 0.5
    sh2 = torch.randn((64,), device="cuda", generator=g) * 0.3
    return cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2


def pixels(torch, shape, seed):
    """uint8 pixels minus the ImageNet-Caffe BGR mean: what preprocess_image feeds (facerec_test.py:93-102)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    p = torch.randint(0, 256, shape, device="cuda", generator=g).float()
    return (p - torch.tensor([103.939, 116.779, 123.68], device="cuda")).contiguous()


def act6(v):
    return np.minimum(np.maximum(v, 0), 6)


@pytest.mark.parametrize("n,h,w", [(2, 192, 192), (1, 224, 224), (3, 96, 96), (1, 100, 100), (2, 13, 21), (1, 7, 5), (1, 3, 3), (2, 33, 64),
                                   (1, 64, 191)])
def test_bounded_stem_vs_exact_stem_and_oracle(env, n, h, w):
    """Every size class: even / odd maps (both paddings of both stride-2 layers), partial patches, maps smaller than a
    patch, n = 1 (the window loads of the last rows run to the very end of the input buffer)."""
    torch, ops = env
    cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2 = weights(torch, h * 3 + w)
    x = pixels(torch, (n, h, w, 3), h + w)
    prep = ops.split_weights_device(kp, x.device)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    y3 = ops.stem3_fused(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep, overflow=flag)
    y2 = ops.stem2_fused(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    assert tuple(y3.shape) == tuple(y2.shape) and int(flag.item()) == 0
    # four fused layers of ReLU6-bounded values, conv1 channels with 40x weights in front: both kernels carry fp32-grade
    # round-off relative to the sum of |terms| of conv1 (which is ~40 x the [0, 6] result there): 1e-5 of the value range
    assert float((y3 - y2).abs().max()) < 6e-5
    if n * h * w <= 3 * 100 * 100:
        xn = x.cpu().numpy().astype(np.float64)
        c1 = act6(tfo.conv2d(xn, cw.cpu().numpy().astype(np.float64), (2, 2), "SAME") + csh.cpu().numpy())
        d1 = act6(tfo.depthwise_conv2d(c1, k1.cpu().numpy()[..., None].astype(np.float64), (1, 1), "SAME") * sc1.cpu().numpy() + sh1.cpu().numpy())
        p1 = act6(d1.reshape(-1, 32).dot(kp.T.astype(np.float64)) + psh.cpu().numpy()).reshape(d1.shape[:3] + (64,))
        want = act6(tfo.depthwise_conv2d(p1, k2.cpu().numpy()[..., None].astype(np.float64), (2, 2), "SAME") * sc2.cpu().numpy() + sh2.cpu().numpy())
        e3, e2 = float(np.abs(y3.cpu().numpy() - want).max()), float(np.abs(y2.cpu().numpy() - want).max())
        assert e3 < 6e-5 and e3 < 4 * e2 + 2e-6, (e3, e2)          # within 4x of the exact-fp32-conv1 kernel's own distance to fp64 (the pointwise tests' rule)


def test_bounded_stem_full_size_every_element_and_run_to_run(env):
    """Batch 256 @ 192x192 (the BASELINE workload): every output element against the unfused kernels (exact fp32 conv1), and
    three launches bit-identical -- 36 patches per persistent workgroup, the window of the next patch in flight, LDS regions
    re-used across stages."""
    torch, ops = env
    cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2 = weights(torch, 7)
    x = pixels(torch, (256, 192, 192, 3), 3)
    prep = ops.split_weights_device(kp, x.device)
    d2 = ops.dwconv3x3(ops.pwconv1x1_f16split(ops.dwconv3x3(ops.conv3x3_c3(x, cw, csh, 2), k1, sc1, sh1, 1), None, psh, prepared=prep), k2, sc2, sh2, 2)
    ys = [ops.stem3_fused(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep) for _ in range(3)]
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    assert float((ys[0] - d2).abs().max()) < 6e-5
    # the synthetic workload of bench.py (U(-128, 128)) respects the bound too
    g = torch.Generator(device="cuda").manual_seed(5)
    xu = (torch.rand((64, 192, 192, 3), device="cuda", generator=g) * 256 - 128).contiguous()
    y3 = ops.stem3_fused(xu, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    y2 = ops.stem2_fused(xu, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    assert float((y3 - y2).abs().max()) < 6e-5


def test_the_declared_bound_is_checked_on_the_device(env):
    torch, ops = env
    cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2 = weights(torch, 11)
    x = pixels(torch, (2, 48, 48, 3), 1)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.stem3_fused(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2, overflow=flag)
    assert int(flag.item()) == 0
    x[1, 20, 31, 2] = 256.0                                  # the bound is |x| < 256 (in_log2 = 7)
    ops.stem3_fused(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2, overflow=flag)
    assert int(flag.item()) == 1
    flag.zero_()
    x[1, 20, 31, 2] = float("nan")
    ops.stem3_fused(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2, overflow=flag)
    assert int(flag.item()) == 1
    # a wider bound (in_log2 = 5: |x| < 1024) takes the same values without complaint and agrees with the exact kernel
    flag.zero_()
    x[1, 20, 31, 2] = 700.0
    y3 = ops.stem3_fused(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2, in_log2=5, overflow=flag)
    assert int(flag.item()) == 0
    assert float((y3 - ops.stem2_fused(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2)).abs().max()) < 6e-5


def test_engine_uses_the_bound_and_reports_violations(env):
    """lower_graph(input_bound=256) -> wire kind 17; same embeddings as the exact-conv1 plan to round-off; a batch that breaks
    the bound raises the engine's flag (and the NumPy entry of the drop-in class refuses it up front)."""
    torch, ops = env
    from hse_facerec_tf_amd import TensorFlowInference, graphdef, lowering
    from hse_facerec_tf_amd.engine import Engine
    from conftest import MODEL_PB
    g = graphdef.read_graph(MODEL_PB)
    pa = lowering.lower_graph(g, "input_1:0", {0: "global_pooling/Mean:0"}, (192, 192), input_bound=256.0)
    pb = lowering.lower_graph(g, "input_1:0", {0: "global_pooling/Mean:0"}, (192, 192))
    assert pa.layers[0].kind == lowering.OP_STEM3_F16S and pa.layers[0].in_log2 == 7 and pb.layers[0].kind == lowering.OP_STEM2_F16S
    x = pixels(torch, (4, 192, 192, 3), 9)
    ea, eb = Engine(pa, max_batch=4), Engine(pb, max_batch=4)
    fa, fb = ea.forward(x)["features"], eb.forward(x)["features"]
    assert float((fa - fb).abs().max() / fb.abs().max()) < 1e-5
    assert ea.input_overflow() is False
    x[2, 5, 5, 0] = 300.0
    ea.forward(x)
    assert ea.input_overflow() is True and ea.input_overflow() is False          # read once, then cleared
    ea.close(), eb.close()
    tfi = TensorFlowInference(MODEL_PB, 'input_1:0', 'global_pooling/Mean:0', input_size=(96, 96), max_batch=4)
    assert tfi.plan.layers[0].kind == lowering.OP_STEM3_F16S
    with pytest.raises(ValueError):
        tfi.extract_batch(np.full((1, 96, 96, 3), 300.0, np.float32))
    exact = TensorFlowInference(MODEL_PB, 'input_1:0', 'global_pooling/Mean:0', input_size=(96, 96), max_batch=4, input_bound=None)
    assert exact.plan.layers[0].kind == lowering.OP_STEM2_F16S
    assert exact.extract_batch(np.full((1, 96, 96, 3), 300.0, np.float32)).shape == (1, 1024)
    tfi.close_session(), exact.close_session()
