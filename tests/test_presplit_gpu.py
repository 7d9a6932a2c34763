"""GPU parity of the pre-split activation path (round 2): the depthwise kernel storing its result as split rows
(csrc/dwconv.hip SPLIT) and the split-f16 GEMM that stages both operands by LDS-DMA (csrc/pwconv_ps.hip).  Same graph
nodes and the same bar as the kernels they stand in for (facerec_test.py:120 / facial_analysis.py:109)."""
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


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("n,h,w,c,stride", [(2, 12, 12, 512, 1), (1, 24, 24, 256, 2), (3, 7, 5, 32, 1), (2, 9, 11, 64, 2), (1, 1, 1, 1024, 1),
                                            (256, 12, 12, 512, 1), (256, 48, 48, 128, 2)])
def test_depthwise_split_rows_are_the_split_of_the_fp32_result(env, n, h, w, c, stride):
    """Bit for bit: hi = f16(v * 2^12), lo = f16(v * 2^12 - hi) of what dwconv3x3 writes as fp32 -- every element, both strides,
    odd sizes, the BASELINE sizes last (lane-pair exchange across every 32-channel group of every pixel)."""
    torch, ops = env
    g = torch.Generator(device="cuda").manual_seed(h * 13 + c + stride)
    x = torch.rand((n, h, w, c), device="cuda", generator=g) * 6
    x[torch.rand((n, h, w, c), device="cuda", generator=g) < 0.3] = 0.0
    kd = torch.randn((3, 3, c), device="cuda", generator=g) / 3
    sc = torch.rand((c,), device="cuda", generator=g) + 0.5
    sh = torch.randn((c,), device="cuda", generator=g) * 0.3
    y = ops.dwconv3x3(x, kd, sc, sh, stride)
    ys = ops.dwconv3x3_split(x, kd, sc, sh, stride)
    want = ops.split_rows_encode(y)
    assert ys.shape == want.shape and ys.dtype == torch.float16
    assert torch.equal(ys.view(torch.int16), want.view(torch.int16))
    # and the pair (hi, lo) stands for the fp32 value to 2^-22
    back = ops.split_rows_decode(ys)
    assert float((back - y).abs().max()) <= float(y.abs().max()) * 2.0 ** -21


def test_depthwise_split_rejects_what_the_format_cannot_hold(env):
    torch, ops = env
    z = lambda *s: torch.zeros(s, device="cuda")
    with pytest.raises(NotImplementedError):
        ops.dwconv3x3_split(z(1, 4, 4, 48), z(3, 3, 48), z(48), z(48))            # c % 32 != 0
    with pytest.raises(ValueError):
        ops.dwconv3x3_split(z(1, 4, 4, 32), z(3, 3, 32), z(32), z(32), act=1)     # plain ReLU: no bound
    with pytest.raises(ValueError):
        ops.dwconv3x3_split(z(1, 4, 4, 32), z(3, 3, 32), z(32), z(32), a_log2=13)


@pytest.mark.parametrize("m,k,cout", [(48 * 48 * 2, 64, 128), (2304, 128, 128), (1152 + 7, 128, 256), (576, 256, 256), (300, 256, 512),
                                      (36 * 5, 512, 512), (36 * 3 + 1, 512, 1024), (129, 1024, 1024), (1, 1024, 1024), (288 * 3 + 17, 32, 128),
                                      (255, 96, 384)])
def test_presplit_gemm_vs_oracle(env, m, k, cout):
    """fp32-grade results (the bar of the fp32-MFMA kernel) on ReLU6-range inputs with exact zeros, the bound itself, tiny
    values, and output channels six decades apart; ragged M (tail tiles), every K from one step to 32."""
    torch, ops = env
    rs = np.random.RandomState(m + k + cout + 2)
    x = rs.uniform(0, 6, (m, k)).astype(np.float32)
    x[rs.rand(m, k) < 0.3] = 0.0
    x[rs.rand(m, k) < 0.05] = 6.0
    x[rs.rand(m, k) < 0.05] *= 1e-5
    w1 = (rs.randn(k, cout) / np.sqrt(k)).astype(np.float32)
    w1[:, cout // 2] = 0.0
    sh = rs.randn(cout).astype(np.float32)
    tol = TOL * max(1.0, (k / 256.0) ** 0.5)
    xd = torch.from_numpy(x).cuda()
    xs = ops.split_rows_encode(xd)
    shd = torch.from_numpy(sh).cuda()
    y = ops.pwconv1x1_presplit(xs, w1.T, shd, 2).cpu().numpy()
    assert rel(y, np.minimum(np.maximum(x.astype(np.float64).dot(w1.astype(np.float64)) + sh, 0), 6)) < tol
    w2 = w1 * (10.0 ** rs.uniform(-3, 3, cout)).astype(np.float32)[None, :]
    want = x.astype(np.float64).dot(w2.astype(np.float64)) + sh
    y = ops.pwconv1x1_presplit(xs, w2.T, shd, 0).cpu().numpy()
    y32 = ops.pwconv1x1(xd, torch.from_numpy(np.ascontiguousarray(w2.T)).cuda(), shd, 0).cpu().numpy() if cout % 64 == 0 and k % 32 == 0 else None
    scale = np.abs(x.astype(np.float64)).dot(np.abs(w2.astype(np.float64))).max(axis=0) + np.abs(sh)
    err = (np.abs(y - want) / scale).max(axis=0)
    assert err.max() < tol, "channel %d" % err.argmax()
    if y32 is not None:
        err32 = (np.abs(y32 - want) / scale).max(axis=0)
        assert err.max() < 4 * max(err32.max(), 2.0 ** -24), "pre-split %g vs fp32 MFMA %g" % (err.max(), err32.max())
    # and it agrees with the register-staged split-f16 GEMM to round-off (same products, another summation order)
    y16 = ops.pwconv1x1_f16split(xd, w2.T, shd, 0).cpu().numpy() if cout % 64 == 0 else None
    if y16 is not None:
        assert (np.abs(y - y16) / scale).max() < tol


def test_presplit_gemm_operand_maps_with_exact_integers(env):
    """Selector rows against an asymmetric integer kernel: exact in the split, so a row/column, k-slot or swizzle mix-up in
    the DMA source permutation or the 16x16x32 fragment maps gives a wrong INTEGER."""
    torch, ops = env
    m, k, cout = 288 + 40, 64, 256
    x = np.zeros((m, k), np.float32)
    for i in range(m):
        x[i, (7 * i + 3) % k] = 1 + (i % 5)
    w = (np.arange(cout)[:, None] * 3 - np.arange(k)[None, :] * 5 + 11).astype(np.float32) % 17 - 8      # [cout, k], asymmetric
    want = x.astype(np.float64).dot(w.T.astype(np.float64))
    xs = ops.split_rows_encode(torch.from_numpy(x).cuda())
    y = ops.pwconv1x1_presplit(xs, w, torch.zeros(cout, device="cuda"), 0).cpu().numpy()
    assert np.array_equal(y, want.astype(np.float32))


@pytest.mark.parametrize("m,k,cout", [(36864, 512, 512), (9216, 1024, 1024), (9216, 512, 1024), (147456, 128, 256), (36864 + 77, 256, 512),
                                      (25088, 1024, 1024)])
def test_presplit_gemm_full_size_every_element_and_run_to_run(env, m, k, cout):
    """BASELINE-size GEMMs (batch 256 at 192, and the 7x7 map of batch 512 at 224): persistent workgroups walk several
    tiles each, the DMA ring runs two steps ahead across tile boundaries, the epilogue borrows the released ring slot.
    EVERY output element against fp64 (on the device); three launches must agree bit for bit."""
    torch, ops = env
    g = torch.Generator(device="cuda").manual_seed(m + k)
    x = torch.rand((m, k), device="cuda", generator=g) * 6
    x[torch.rand((m, k), device="cuda", generator=g) < 0.2] = 0.0
    w = torch.randn((cout, k), device="cuda", generator=g) / k ** 0.5
    sh = torch.randn((cout,), device="cuda", generator=g)
    want = torch.clamp(x.double() @ w.double().T + sh.double(), 0, 6)
    prep = ops.split_weights_device(w, x.device)
    xs = ops.split_rows_encode(x)
    ys = [ops.pwconv1x1_presplit(xs, None, sh, prepared=prep) for _ in range(3)]
    err = float((ys[0].double() - want).abs().max() / want.abs().max())
    assert err < TOL * max(1.0, (k / 256.0) ** 0.5), err
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    y16 = ops.pwconv1x1_f16split(x, None, sh, prepared=prep)
    err16 = float((y16.double() - want).abs().max() / want.abs().max())
    assert err < 1.5 * err16 + 1e-7


def test_presplit_gemm_rejects_uncovered_shapes(env):
    torch, ops = env
    xs = torch.zeros((4, 2, 2, 32), device="cuda", dtype=torch.float16)
    with pytest.raises(NotImplementedError):
        ops.pwconv1x1_presplit(xs, np.zeros((64, 64), np.float32), torch.zeros(64, device="cuda"))        # cout % 128 != 0
    assert ops.pwconv1x1_presplit(xs[:0], np.zeros((128, 64), np.float32), torch.zeros(128, device="cuda")).shape == (0, 128)


def test_whole_network_presplit_vs_fp32_tensors(env):
    """The engine with pre-split activations (default) against the same plan with fp32 tensors and the register-staged
    GEMM: embeddings agree to round-off, and the lowering really converted the eight deep (K >= 256) depthwise -> pointwise
    tensors."""
    torch, ops = env
    from hse_facerec_tf_amd import graphdef, lowering
    from hse_facerec_tf_amd.engine import Engine
    from conftest import MODEL_PB
    g = graphdef.read_graph(MODEL_PB)
    fetch = {0: "global_pooling/Mean:0", 1: "age_pred/Softmax:0", 2: "gender_pred/Sigmoid:0"}
    pa = lowering.lower_graph(g, "input_1:0", fetch, (192, 192))
    pb = lowering.lower_graph(g, "input_1:0", fetch, (192, 192), presplit="none")
    assert sum(1 for L in pa.layers if L.in_split) == 8 and sum(1 for L in pa.layers if L.out_split) == 8
    assert not any(L.in_split or L.out_split for L in pb.layers)
    x = torch.from_numpy(np.random.RandomState(5).uniform(-128, 128, (5, 192, 192, 3)).astype(np.float32)).cuda()
    ea, eb = Engine(pa, max_batch=5), Engine(pb, max_batch=5)
    ra, rb = ea.forward(x, (0, 1, 2)), eb.forward(x, (0, 1, 2))
    for k in ("features", "age_probs", "gender"):
        a, b = ra[k].cpu().numpy(), rb[k].cpu().numpy()
        assert rel(a, b) < 1e-5, k
    ea.close(), eb.close()


@pytest.mark.parametrize("n,h,w,k,cout,act,s", [(2, 12, 12, 512, 512, 2, 1), (5, 12, 12, 256, 512, 2, 1), (8, 6, 6, 512, 1024, 2, 1), (19, 6, 6, 1024, 1024, 2, 1),
                                                (3, 3, 3, 256, 128, 2, 1), (4, 4, 3, 256, 256, 1, 1), (1, 12, 12, 288, 384, 0, 1),
                                                (2, 12, 12, 512, 512, 2, 2), (7, 12, 12, 256, 384, 2, 2), (5, 14, 14, 512, 512, 2, 2),
                                                (3, 14, 14, 512, 512, 2, 1), (11, 7, 7, 1024, 1024, 2, 1), (4, 10, 10, 256, 256, 2, 1), (2, 16, 18, 256, 128, 2, 1)])
def test_pointwise_with_depthwise_epilogue_vs_oracle(n, h, w, k, cout, act, s):
    """csrc/pwconv_ps.hip with DW = true: pointwise on split rows + the next block's depthwise 3x3 / 1 / SAME + scale + shift + ReLU6
    in the epilogue, output as split rows -- against the fp64 oracle of the two layers, at the pre-split GEMM's 2e-6 bar
    (relative to the output scale).  Shapes: MobileNet's 12x12x512 and 6x6x1024 blocks, odd image counts (tiles with fewer
    maps than they hold: rows beyond M), a 3x3 map, a non-square map, K and Cout that are not powers of two, and the stride-2
    depthwise of the 12x12 -> 6x6 block."""
    import torch
    from hse_facerec_tf_amd import ops
    from oracle import tf_graph as tfo
    rs = np.random.RandomState(n * 31 + h + k)
    x = rs.uniform(0, 6, (n, h, w, k)).astype(np.float32)
    x[rs.uniform(size=x.shape) < 0.3] = 0.0
    wt = (rs.randn(cout, k) / np.sqrt(k)).astype(np.float32)
    sh = rs.randn(cout).astype(np.float32)
    dww = (rs.randn(3, 3, cout) * 0.4).astype(np.float32)
    dsc = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    dsh = rs.randn(cout).astype(np.float32)
    mid = x.reshape(-1, k).astype(np.float64).dot(wt.T.astype(np.float64)) + sh
    mid = np.minimum(np.maximum(mid, 0), 6) if act == 2 else (np.maximum(mid, 0) if act == 1 else mid)
    mid = mid.reshape(n, h, w, cout)
    pad0 = 1 if s == 1 else 0          # TF SAME: stride 2 on an even map pads bottom / right only
    want = tfo.depthwise_conv2d(np.pad(mid, ((0, 0), (pad0, 1), (pad0, 1), (0, 0))), dww.reshape(3, 3, cout, 1).astype(np.float64), (s, s), "VALID")
    want = np.minimum(np.maximum(want * dsc + dsh, 0), 6)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ys = ops.pwconv1x1_presplit_dw(ops.split_rows_encode(d(x), 12), wt, d(sh), d(dww), d(dsc), d(dsh), act, 12, 12, dw_stride=s)
    assert tuple(ys.shape) == (n, h // s, w // s, cout // 32, 2, 32)
    got = ops.split_rows_decode(ys, 12).cpu().numpy()
    # error scale: the depthwise sums up to 9 pointwise results of magnitude |mid| (<= 6 with ReLU6) times |tap| * scale
    scale = np.abs(dww).sum(axis=(0, 1)) * dsc * max(np.abs(mid).max(), 1.0) + np.abs(dsh) + 1.0
    assert (np.abs(got - want) / scale).max() < 2e-6
    # the same numbers as the unfused pair on the device (pre-split GEMM, then the split-row depthwise kernel), up to the depthwise's summation order
    mid_d = ops.pwconv1x1_presplit(ops.split_rows_encode(d(x), 12), wt, d(sh), act, 12)
    two = ops.split_rows_decode(ops.dwconv3x3_split(mid_d, d(dww), d(dsc), d(dsh), s, 2, 12), 12).cpu().numpy()
    assert (np.abs(got - two) / scale).max() < 1e-6
    with pytest.raises(Exception):            # a 17x17 map does not fit a 288-row tile
        ops.pwconv1x1_presplit_dw(ops.split_rows_encode(d(rs.uniform(0, 6, (1, 17, 17, k)).astype(np.float32)), 12), wt, d(sh), d(dww), d(dsc), d(dsh),
                                  act, 12, 12)


@pytest.mark.parametrize("n,h,w,k,cout,act", [(8, 6, 6, 1024, 1024, 2), (19, 6, 6, 512, 384, 2), (3, 12, 12, 256, 128, 1), (5, 6, 12, 288, 256, 0),
                                              (12, 7, 7, 1024, 1024, 2), (2, 14, 14, 256, 128, 2)])
def test_pointwise_with_global_pool_epilogue_vs_oracle(n, h, w, k, cout, act):
    """csrc/pwconv_ps.hip, epilogue mode 4: pointwise on split rows + the global average pool, against the fp64 oracle and against
    the unfused pair on the device (maps of 36, 72 and 144 pixels; image counts that leave the last tile partly empty)."""
    import torch
    from hse_facerec_tf_amd import ops
    rs = np.random.RandomState(n + h * 7 + k)
    x = rs.uniform(0, 6, (n, h, w, k)).astype(np.float32)
    wt = (rs.randn(cout, k) / np.sqrt(k)).astype(np.float32)
    sh = rs.randn(cout).astype(np.float32)
    mid = x.reshape(-1, k).astype(np.float64).dot(wt.T.astype(np.float64)) + sh
    mid = np.minimum(np.maximum(mid, 0), 6) if act == 2 else (np.maximum(mid, 0) if act == 1 else mid)
    want = mid.reshape(n, h * w, cout).mean(axis=1)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    xs = ops.split_rows_encode(d(x), 12)
    got = ops.pwconv1x1_presplit_gap(xs, wt, d(sh), act, 12)
    assert tuple(got.shape) == (n, cout)
    assert np.abs(got.cpu().numpy() - want).max() < 2e-6 * max(np.abs(mid).max(), 1.0)
    two = ops.gap(ops.pwconv1x1_presplit(xs, wt, d(sh), act, 12)).reshape(n, cout)
    assert float((got - two).abs().max()) < 1e-6 * max(np.abs(mid).max(), 1.0)
    with pytest.raises(Exception):            # 3x3 maps: more than eight per tile
        ops.pwconv1x1_presplit_gap(ops.split_rows_encode(d(np.ascontiguousarray(x[:, :3, :3])), 12), wt, d(sh), act, 12)
