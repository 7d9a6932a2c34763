"""GPU parity of the fourth-generation fused stem (csrc/stem4_fused.hip, round 3): graph nodes #30-#55 (facerec_test.py:120 /
facial_analysis.py:109) for inputs whose edges are multiples of 4 -- the window converted to f16 once, conv1's MFMA operands
read straight from it (no im2col) -- against the kernel it replaces (stem3), the exact-fp32-conv1 kernel (stem2) and the fp64
oracle; and its uint8 form (the resized RGB bytes in, float conversion + channel reversal + mean of facerec_test.py:95-106
folded into the constants) against the fp32 form on the preprocessed floats of the same bytes."""
import numpy as np
import pytest

from oracle import tf_graph as tfo
from test_stem3_gpu import act6, pixels, weights

pytestmark = pytest.mark.gpu
MEAN_BGR = (103.939, 116.779, 123.68)


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    from hse_facerec_tf_amd import ops
    return torch, ops


def oracle_stem(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2):
    xn = x.astype(np.float64)
    c1 = act6(tfo.conv2d(xn, cw.cpu().numpy().astype(np.float64), (2, 2), "SAME") + csh.cpu().numpy())
    d1 = act6(tfo.depthwise_conv2d(c1, k1.cpu().numpy()[..., None].astype(np.float64), (1, 1), "SAME") * sc1.cpu().numpy() + sh1.cpu().numpy())
    p1 = act6(d1.reshape(-1, 32).dot(kp.T.astype(np.float64)) + psh.cpu().numpy()).reshape(d1.shape[:3] + (64,))
    return act6(tfo.depthwise_conv2d(p1, k2.cpu().numpy()[..., None].astype(np.float64), (2, 2), "SAME") * sc2.cpu().numpy() + sh2.cpu().numpy())


@pytest.mark.parametrize("n,h,w", [(2, 192, 192), (1, 224, 224), (3, 96, 96), (1, 100, 100), (2, 12, 20), (1, 4, 4), (1, 8, 4), (2, 32, 64),
                                   (1, 64, 188), (5, 36, 44)])
def test_stem4_vs_stem3_stem2_and_oracle(env, n, h, w):
    """Interior and border patches, partial patches (100 -> 25 outputs: the last patch hangs over the map), maps smaller than a
    patch, n = 1 (window pieces of the last rows run to the very end of the input buffer), several images per workgroup."""
    torch, ops = env
    cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2 = weights(torch, h * 3 + w)
    x = pixels(torch, (n, h, w, 3), h + w)
    prep = ops.split_weights_device(kp, x.device)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    y4 = ops.stem4_fused(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep, overflow=flag)
    y3 = ops.stem3_fused(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    y2 = ops.stem2_fused(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    assert tuple(y4.shape) == tuple(y2.shape) == (n, h // 4, w // 4, 64) and int(flag.item()) == 0
    assert float((y4 - y3).abs().max()) < 6e-5 and float((y4 - y2).abs().max()) < 6e-5
    if n * h * w <= 3 * 100 * 100:
        want = oracle_stem(x.cpu().numpy(), cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2)
        e4, e2 = float(np.abs(y4.cpu().numpy() - want).max()), float(np.abs(y2.cpu().numpy() - want).max())
        assert e4 < 6e-5 and e4 < 4 * e2 + 2e-6, (e4, e2)      # within 4x of the exact-fp32-conv1 kernel's own distance to fp64


@pytest.mark.parametrize("n,h,w", [(2, 192, 192), (1, 224, 224), (3, 96, 96), (1, 100, 100), (2, 12, 20), (1, 4, 4), (2, 32, 64), (5, 36, 44)])
def test_stem4_uint8_input_equals_the_float_path_on_the_same_bytes(env, n, h, w):
    """The RGB bytes in, against (a) the fp32 form of the same kernel on float32(bytes reversed - mean) -- what
    preprocess_image feeds (facerec_test.py:95-106) -- and (b) the fp64 oracle on those floats.  Includes the last conv row /
    column, where a tap on the SAME padding drops out of the folded mean term."""
    torch, ops = env
    cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2 = weights(torch, h * 5 + w)
    g = torch.Generator(device="cuda").manual_seed(h * 7 + w)
    rgb = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    rgb[0, :2] = 255                                            # saturated rows at the image's edge
    rgb[-1, -2:, -3:] = 0
    x = (rgb.flip(-1).double() - torch.tensor(MEAN_BGR, dtype=torch.float64, device="cuda")).float().contiguous()
    prep = ops.split_weights_device(kp, x.device)
    y8 = ops.stem4_fused(rgb, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep, u8_mean_bgr=MEAN_BGR)
    y4 = ops.stem4_fused(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    y2 = ops.stem2_fused(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    assert float((y8 - y4).abs().max()) < 6e-5 and float((y8 - y2).abs().max()) < 6e-5
    if n * h * w <= 3 * 100 * 100:
        want = oracle_stem(x.cpu().numpy(), cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2)
        e8, e2 = float(np.abs(y8.cpu().numpy() - want).max()), float(np.abs(y2.cpu().numpy() - want).max())
        assert e8 < 6e-5 and e8 < 4 * e2 + 2e-6, (e8, e2)


def test_stem4_full_size_every_element_and_run_to_run(env):
    """Batch 256 @ 192x192 (the BASELINE workload): every output element against the unfused kernels (exact fp32 conv1), three
    launches bit-identical, both input forms; the bench's synthetic U(-128, 128) batch as well."""
    torch, ops = env
    cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2 = weights(torch, 7)
    g = torch.Generator(device="cuda").manual_seed(3)
    rgb = torch.randint(0, 256, (256, 192, 192, 3), dtype=torch.uint8, device="cuda", generator=g)
    x = (rgb.flip(-1).double() - torch.tensor(MEAN_BGR, dtype=torch.float64, device="cuda")).float().contiguous()
    prep = ops.split_weights_device(kp, x.device)
    d2 = ops.dwconv3x3(ops.pwconv1x1_f16split(ops.dwconv3x3(ops.conv3x3_c3(x, cw, csh, 2), k1, sc1, sh1, 1), None, psh, prepared=prep), k2, sc2, sh2, 2)
    ys = [ops.stem4_fused(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep) for _ in range(3)]
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    assert float((ys[0] - d2).abs().max()) < 6e-5
    y8 = [ops.stem4_fused(rgb, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep, u8_mean_bgr=MEAN_BGR) for _ in range(3)]
    assert torch.equal(y8[0], y8[1]) and torch.equal(y8[0], y8[2])
    assert float((y8[0] - d2).abs().max()) < 6e-5
    del ys, y8, d2
    gg = torch.Generator(device="cuda").manual_seed(5)
    xu = (torch.rand((64, 192, 192, 3), device="cuda", generator=gg) * 256 - 128).contiguous()
    y4 = ops.stem4_fused(xu, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    y2 = ops.stem2_fused(xu, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    assert float((y4 - y2).abs().max()) < 6e-5


def test_stem4_checks_the_declared_bound_and_rejects_other_shapes(env):
    torch, ops = env
    cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2 = weights(torch, 11)
    x = pixels(torch, (2, 48, 48, 3), 1)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.stem4_fused(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2, overflow=flag)
    assert int(flag.item()) == 0
    x[1, 20, 31, 2] = 256.0                                  # the bound is |x| < 256 (in_log2 = 7)
    ops.stem4_fused(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2, overflow=flag)
    assert int(flag.item()) == 1
    flag.zero_()
    x[1, 20, 31, 2] = float("nan")
    ops.stem4_fused(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2, overflow=flag)
    assert int(flag.item()) == 1
    flag.zero_()
    x[1, 20, 31, 2] = -float("inf")
    ops.stem4_fused(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2, overflow=flag)
    assert int(flag.item()) == 1
    flag.zero_()
    x[1, 20, 31, 2] = 700.0                                  # a wider bound (in_log2 = 5: |x| < 1024) takes it
    y4 = ops.stem4_fused(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2, in_log2=5, overflow=flag)
    assert int(flag.item()) == 0
    assert float((y4 - ops.stem2_fused(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2)).abs().max()) < 6e-5
    for hw in ((50, 48), (48, 46), (33, 33)):               # edges that are not multiples of 4 belong to stem3
        with pytest.raises(ValueError):
            ops.stem4_fused(pixels(torch, (1,) + hw + (3,), 2), cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2)


def test_engine_takes_resized_bytes_and_matches_the_float_entry(env):
    """lower_graph(u8_mean_bgr=...) -> Engine.forward_u8(resized RGB bytes) == Engine.forward(float32(bytes reversed - mean))
    to round-off, at 192 and 224 with three outputs; a plan without the mean, or with edges that are not multiples of 4, says
    so; TensorFlowInference.extract_images runs through the uint8 entry and still equals the per-image reference path."""
    torch, ops = env
    from hse_facerec_tf_amd import graphdef, lowering
    from hse_facerec_tf_amd.engine import Engine
    from conftest import MODEL_PB
    g = graphdef.read_graph(MODEL_PB)
    fetch = {0: "global_pooling/Mean:0", 1: "age_pred/Softmax:0", 2: "gender_pred/Sigmoid:0"}
    for size in (192, 224):
        plan = lowering.lower_graph(g, "input_1:0", fetch, (size, size), input_bound=256.0, u8_mean_bgr=MEAN_BGR)
        eng = Engine(plan, max_batch=6)
        assert eng.accepts_u8
        gen = torch.Generator(device="cuda").manual_seed(size)
        rgb = torch.randint(0, 256, (6, size, size, 3), dtype=torch.uint8, device="cuda", generator=gen)
        x = (rgb.flip(-1).double() - torch.tensor(MEAN_BGR, dtype=torch.float64, device="cuda")).float().contiguous()
        a, b = eng.forward_u8(rgb, (0, 1, 2)), eng.forward(x, (0, 1, 2))
        for k in ("features", "age_probs", "gender"):
            assert float((a[k] - b[k]).abs().max() / b[k].abs().max()) < 2e-6, k
        assert not eng.input_overflow()
        eng.close()
    # both entries against the fp64 oracle of the whole graph (96 x 96 keeps the NumPy interpreter to seconds): the bytes-in
    # entry must be as close to it as the floats-in entry (2e-6 of the feature scale measured; the north-star bar is 1e-4)
    from oracle.tf_graph import GraphOracle
    plan = lowering.lower_graph(g, "input_1:0", {0: fetch[0]}, (96, 96), input_bound=256.0, u8_mean_bgr=MEAN_BGR)
    eng = Engine(plan, max_batch=3)
    gen = torch.Generator(device="cuda").manual_seed(96)
    rgb = torch.randint(0, 256, (3, 96, 96, 3), dtype=torch.uint8, device="cuda", generator=gen)
    rgb[2] = torch.randint(0, 256, (1, 1, 3), dtype=torch.uint8, device="cuda", generator=gen) // 2 + rgb[2] // 2      # a lower-contrast image
    x = (rgb.flip(-1).double() - torch.tensor(MEAN_BGR, dtype=torch.float64, device="cuda")).float().contiguous()
    want = GraphOracle(MODEL_PB, np.float64).run(fetch[0], {"input_1:0": x.cpu().numpy()}).reshape(3, -1)
    f8 = eng.forward_u8(rgb)["features"].cpu().numpy()
    f32 = eng.forward(x)["features"].cpu().numpy()
    scale = np.abs(want).max()
    e8, e32 = float(np.abs(f8 - want).max() / scale), float(np.abs(f32 - want).max() / scale)
    print("features vs the fp64 oracle: bytes-in entry %.2e, floats-in entry %.2e" % (e8, e32))
    assert e8 < 1e-5 and e32 < 1e-5 and e8 < 3 * e32 + 2e-6
    eng.close()
    plain = Engine(lowering.lower_graph(g, "input_1:0", {0: fetch[0]}, (96, 96), input_bound=256.0), max_batch=2)
    odd = Engine(lowering.lower_graph(g, "input_1:0", {0: fetch[0]}, (98, 98), input_bound=256.0, u8_mean_bgr=MEAN_BGR), max_batch=2)
    assert not plain.accepts_u8 and not odd.accepts_u8
    with pytest.raises(NotImplementedError):
        plain.forward_u8(torch.zeros((1, 96, 96, 3), dtype=torch.uint8, device="cuda"))
    plain.close(), odd.close()
