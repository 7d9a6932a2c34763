"""GPU parity of the streaming stem (csrc/stem5_stream.hip, round 4): graph nodes #30-#55 (facerec_test.py:120 /
facial_analysis.py:109) with a wave walking down a 4-column strip, every depthwise output accumulated in registers as its
input rows arrive.  Same MFMA layouts, same product order, same depthwise chains as stem4_fused.hip: the results must be
the SAME BITS as the patch kernel's, in both input forms, on every shape (strips that hang over the map, maps narrower than
a strip, one image, vertical segments), and within round-off of the exact-fp32-conv1 kernel and the fp64 oracle."""
import numpy as np
import pytest

from test_stem3_gpu import pixels, weights
from test_stem4_gpu import MEAN_BGR, oracle_stem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    from hse_facerec_tf_amd import ops
    return torch, ops


@pytest.mark.parametrize("n,h,w", [(2, 192, 192), (1, 224, 224), (3, 96, 96), (1, 100, 100), (2, 12, 20), (1, 4, 4), (1, 8, 4), (2, 32, 64),
                                   (1, 64, 188), (5, 36, 44), (1, 16, 16), (7, 48, 8)])
def test_stem5_equals_stem4_bit_for_bit_and_the_oracle(env, n, h, w):
    torch, ops = env
    cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2 = weights(torch, h * 3 + w)
    x = pixels(torch, (n, h, w, 3), h + w)
    prep = ops.split_weights_device(kp, x.device)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    y5 = ops.stem5_stream(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep, overflow=flag)
    y4 = ops.stem4_fused(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    y2 = ops.stem2_fused(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    assert tuple(y5.shape) == (n, h // 4, w // 4, 64) and int(flag.item()) == 0
    assert bool(torch.isfinite(y5).all())
    d = (y5 - y4).abs()
    assert torch.equal(y5, y4), ("max diff %.3e at %r" % (float(d.max()), tuple(int(v) for v in np.unravel_index(int(d.argmax()), d.shape))))
    assert float((y5 - y2).abs().max()) < 6e-5
    if n * h * w <= 3 * 100 * 100:
        want = oracle_stem(x.cpu().numpy(), cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2)
        e5, e2 = float(np.abs(y5.cpu().numpy() - want).max()), float(np.abs(y2.cpu().numpy() - want).max())
        assert e5 < 6e-5 and e5 < 4 * e2 + 2e-6, (e5, e2)


@pytest.mark.parametrize("n,h,w", [(2, 192, 192), (1, 224, 224), (3, 96, 96), (1, 100, 100), (2, 12, 20), (1, 4, 4), (2, 32, 64), (5, 36, 44)])
def test_stem5_uint8_input(env, n, h, w):
    """The resized RGB bytes in (float conversion, channel reversal and mean of facerec_test.py:95-106 folded into the
    constants): the same bits as stem4's uint8 form, round-off from the float form."""
    torch, ops = env
    cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2 = weights(torch, h * 5 + w)
    g = torch.Generator(device="cuda").manual_seed(h * 7 + w)
    rgb = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    rgb[0, :2] = 255
    rgb[-1, -2:, -3:] = 0
    x = (rgb.flip(-1).double() - torch.tensor(MEAN_BGR, dtype=torch.float64, device="cuda")).float().contiguous()
    prep = ops.split_weights_device(kp, x.device)
    y8 = ops.stem5_stream(rgb, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep, u8_mean_bgr=MEAN_BGR)
    y48 = ops.stem4_fused(rgb, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep, u8_mean_bgr=MEAN_BGR)
    y5 = ops.stem5_stream(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    assert torch.equal(y8, y48)
    assert float((y8 - y5).abs().max()) < 6e-5


def test_stem5_full_size_every_element_and_run_to_run(env):
    """Batch 256 @ 192x192 (the BASELINE workload: one strip per wave, 3072 waves): every element equals the patch kernel's,
    three launches bit-identical, both input forms; batch 512 @ 224 (strips in vertical segments, several units per wave)."""
    torch, ops = env
    cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2 = weights(torch, 7)
    g = torch.Generator(device="cuda").manual_seed(3)
    rgb = torch.randint(0, 256, (256, 192, 192, 3), dtype=torch.uint8, device="cuda", generator=g)
    x = (rgb.flip(-1).double() - torch.tensor(MEAN_BGR, dtype=torch.float64, device="cuda")).float().contiguous()
    prep = ops.split_weights_device(kp, x.device)
    y4 = ops.stem4_fused(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    ys = [ops.stem5_stream(x, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep) for _ in range(3)]
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]) and torch.equal(ys[0], y4)
    y48 = ops.stem4_fused(rgb, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep, u8_mean_bgr=MEAN_BGR)
    y8 = [ops.stem5_stream(rgb, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep, u8_mean_bgr=MEAN_BGR) for _ in range(3)]
    assert torch.equal(y8[0], y8[1]) and torch.equal(y8[0], y8[2]) and torch.equal(y8[0], y48)
    del ys, y8, y4, y48, x, rgb
    gg = torch.Generator(device="cuda").manual_seed(5)
    xu = (torch.rand((160, 224, 224, 3), device="cuda", generator=gg) * 256 - 128).contiguous()
    y5 = ops.stem5_stream(xu, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    y4 = ops.stem4_fused(xu, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep)
    assert torch.equal(y5, y4)


def test_stem5_checks_the_declared_bound_and_rejects_other_shapes(env):
    torch, ops = env
    cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2 = weights(torch, 11)
    x = pixels(torch, (2, 48, 48, 3), 1)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.stem5_stream(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2, overflow=flag)
    assert int(flag.item()) == 0
    for bad in (256.0, float("nan"), -float("inf")):
        x[1, 20, 31, 2] = bad
        ops.stem5_stream(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2, overflow=flag)
        assert int(flag.item()) == 1
        flag.zero_()
    x[1, 20, 31, 2] = 700.0                                  # a wider bound (in_log2 = 5: |x| < 1024) takes it
    y5 = ops.stem5_stream(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2, in_log2=5, overflow=flag)
    assert int(flag.item()) == 0
    assert float((y5 - ops.stem2_fused(x, cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2)).abs().max()) < 6e-5
    for hw in ((50, 48), (48, 46), (33, 33)):
        with pytest.raises(ValueError):
            ops.stem5_stream(pixels(torch, (1,) + hw + (3,), 2), cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2)


def test_stem5_batch_beyond_one_launch_goes_as_ranges_of_images(env):
    """A launch addresses 2 GB of (fp32-sized) input: 11 651 images of 64 x 64 x 3 x 4 B.  A larger batch is split into ranges of
    images by the launcher -- same bits as the ranges run one by one."""
    torch, ops = env
    n, h, w = 12000, 64, 64
    cw, csh, k1, sc1, sh1, kp, psh, k2, sc2, sh2 = weights(torch, 77)
    g = torch.Generator(device="cuda").manual_seed(5)
    rgb = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    prep = ops.split_weights_device(kp, rgb.device)
    y = ops.stem5_stream(rgb, cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep, u8_mean_bgr=MEAN_BGR)
    assert tuple(y.shape) == (n, 16, 16, 64)
    for a, b in ((0, 3000), (11000, 12000), (11600, 11700)):          # ranges on both sides of the launcher's cut at image 11 651
        part = ops.stem5_stream(rgb[a:b].contiguous(), cw, csh, k1, sc1, sh1, None, psh, k2, sc2, sh2, prepared=prep, u8_mean_bgr=MEAN_BGR)
        assert torch.equal(y[a:b], part), (a, b)
