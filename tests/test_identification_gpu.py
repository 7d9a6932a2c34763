"""GPU parity for the identification stage (facerec_test.py:401-432) vs scikit-learn's own
results frozen in tests/golden/nn1.npz, plus a world-size-1 run of the sharded gallery path."""
import os

import numpy as np
import pytest

from oracle import identification as oid

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_l2_normalize_matches_sklearn_fixture(torch_):
    from hse_facerec_tf_amd import ops
    z = np.load(os.path.join(GOLDEN, "nn1.npz"))
    X, y = oid.synthetic_gallery(int(z["n_classes"]), int(z["dim"]), int(z["seed"]), float(z["noise"]))
    Xn = ops.l2_normalize(torch_.from_numpy(X).cuda()).cpu().numpy()
    assert np.abs(Xn[z["kept"][:8]] - z["x_norm_sample"]).max() < 1e-6
    zero = ops.l2_normalize(torch_.zeros((2, 16), device="cuda"))
    assert float(zero.abs().max()) == 0.0                                  # sklearn leaves zero rows alone


def test_one_nn_protocol_matches_sklearn(torch_):
    from hse_facerec_tf_amd import identification
    z = np.load(os.path.join(GOLDEN, "nn1.npz"))
    X, y = oid.synthetic_gallery(int(z["n_classes"]), int(z["dim"]), int(z["seed"]), float(z["noise"]))
    r = identification.one_nn_identification(X, y)
    assert np.array_equal(r["indices"], z["kept"]) and np.array_equal(r["y"], z["y"])
    assert np.array_equal(r["train"], z["train"]) and np.array_equal(r["test"], z["test"])
    # index work is bit-exact: same nearest gallery row for every probe (no fp ties in this fixture)
    assert np.array_equal(r["nn_index"], z["nn_index"])
    assert np.array_equal(r["y_pred"], z["y_pred"])
    assert r["accuracy"] == pytest.approx(float(z["accuracy"]), abs=1e-12)
    assert np.abs(r["nn_dist"] - z["nn_dist"]).max() < 2e-4


@pytest.mark.parametrize("nq,ng,d", [(1, 1, 8), (33, 65, 64), (200, 1000, 1024), (70, 31, 2048)])
def test_nn1_vs_bruteforce(torch_, nq, ng, d):
    from hse_facerec_tf_amd import ops
    rs = np.random.RandomState(nq + ng)
    q = rs.randn(nq, d).astype(np.float32)
    g = rs.randn(ng, d).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    d2 = ((q.astype(np.float64)[:, None, :] - g.astype(np.float64)[None]) ** 2).sum(-1) if nq * ng * d < 5e7 else \
        (2 - 2 * q.astype(np.float64) @ g.astype(np.float64).T)
    idx, dist = ops.nn1(torch_.from_numpy(q).cuda(), torch_.from_numpy(g).cuda())
    idx = idx.cpu().numpy()
    best = d2.min(axis=1)
    assert np.all(d2[np.arange(nq), idx] <= best + 1e-5)       # the chosen row is a nearest one (fp32 tolerance)
    assert (idx == d2.argmin(axis=1)).mean() > 0.99
    assert np.abs(dist.cpu().numpy() - best).max() < 1e-4


@pytest.mark.parametrize("nq,ng,d,scale", [(300, 1000, 1024, 1.0), (513, 2049, 512, 37.5), (257, 4100, 256, 3e-4), (1100, 1000, 288 + 32, 1.0)])
def test_nn1_split_f16_gemm_path_vs_fp64(torch_, nq, ng, d, scale):
    """searches of nq * ng * d >= 2^28 run on the split-f16 GEMM (probes scaled by a power of two into [-1, 1],
    the gallery split row by row, |g|^2 - 2 q.g out of the GEMM's epilogue, a row arg-min kernel): ragged gallery sizes (padded to
    64 columns), UN-normalised rows of any magnitude (the gallery / probe protocol feeds raw features: facerec_test.py:284-285),
    duplicated gallery rows (ties -> the lowest index) and a zero row."""
    from hse_facerec_tf_amd import ops
    assert nq * ng * d >= 1 << 28
    rs = np.random.RandomState(nq + ng + d)
    centres = rs.randn(50, d)
    g = (centres[rs.randint(0, 50, ng)] + 0.7 * rs.randn(ng, d)).astype(np.float32) * np.float32(scale)
    q = (centres[rs.randint(0, 50, nq)] + 0.7 * rs.randn(nq, d)).astype(np.float32) * np.float32(scale)
    g[ng // 2] = g[3]                                        # an exact duplicate: the probe next to it must pick index 3
    q[5] = g[3]
    q[6] = 0.0
    g[7] *= np.float32(1e-3)                                 # rows of very different magnitude get their own power of two
    d2 = (q.astype(np.float64) ** 2).sum(1)[:, None] + (g.astype(np.float64) ** 2).sum(1)[None] - 2 * q.astype(np.float64) @ g.astype(np.float64).T
    idx, dist = ops.nn1(torch_.from_numpy(q).cuda(), torch_.from_numpy(g).cuda())
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    best = d2.min(axis=1)
    unit = float((np.abs(q).max() * np.abs(g).max()) * d) * 2.0 ** -20            # the fp32-grade error scale of a d-term product sum
    assert np.all(d2[np.arange(nq), idx] <= best + unit)     # the chosen row is a nearest one
    assert (idx == d2.argmin(axis=1)).mean() > 0.99 and idx[5] == 3
    assert np.abs(dist - np.maximum(best, 0)).max() <= unit
    assert 0 <= idx.min() and idx.max() < ng


def test_nn1_gemm_path_walks_large_searches_in_query_blocks(torch_):
    """the [nq, ng] distance matrix of the split-f16 GEMM path is bounded (256 MiB slices, walked in query blocks; a failed
    workspace allocation falls back to the workspace-free kernel) instead of growing with the search.  20 000 x 4096 (a 327 MB matrix)
    takes two blocks: the same neighbours as the same search done in halves that fit one block each -- rows are independent -- and
    a sample of rows on both sides of the block boundary against fp64."""
    from hse_facerec_tf_amd import ops
    nq, ng, d = 20000, 4096, 32
    assert nq * ng * d >= 1 << 28 and nq * ng * 4 > 256 << 20
    g_ = torch_.Generator(device="cuda").manual_seed(5)
    gal = torch_.randn((ng, d), device="cuda", generator=g_)
    q = gal[torch_.randint(0, ng, (nq,), device="cuda", generator=g_)] + 0.3 * torch_.randn((nq, d), device="cuda", generator=g_)
    idx, dist = ops.nn1(q, gal)
    halves = [ops.nn1(q[a:b].contiguous(), gal) for a, b in ((0, 10000), (10000, 20000))]
    # (each search scales its probes by ONE power of two taken from its own largest value: the halves may use another exponent than
    # the whole, which changes no product but can move the split's last bit -- indices must agree, distances to rounding)
    assert torch_.equal(idx, torch_.cat([h[0] for h in halves]))
    assert float((dist - torch_.cat([h[1] for h in halves])).abs().max()) < 1e-4
    rows = np.array([0, 1, 9999, 10000, 16383, 16384, 16385, 19999])           # both sides of the block boundary
    d2 = ((q[rows].double()[:, None, :] - gal.double()[None]) ** 2).sum(-1)
    assert torch_.equal(idx[rows].long(), d2.argmin(dim=1))
    assert float((dist[rows].double() - d2.min(dim=1).values).abs().max()) < 1e-3


def test_nn1_ties_resolve_to_lowest_index(torch_):
    from hse_facerec_tf_amd import ops
    g = np.zeros((40, 8), np.float32)
    g[:, 0] = 1.0                                            # 40 identical gallery rows
    q = np.zeros((3, 8), np.float32)
    q[:, 0] = 1.0
    idx, _ = ops.nn1(torch_.from_numpy(q).cuda(), torch_.from_numpy(g).cuda())
    assert idx.cpu().tolist() == [0, 0, 0]


def test_sharded_gallery_world1_on_device(torch_):
    from hse_facerec_tf_amd import TensorFlowInference, gallery
    from conftest import MODEL_PB
    tfi = TensorFlowInference(MODEL_PB, 'input_1:0', 'global_pooling/Mean:0', input_size=(96, 96), max_batch=8)
    rs = np.random.RandomState(4)
    imgs = rs.uniform(-128, 128, (19, 96, 96, 3)).astype(np.float32)

    def extract(ids):
        return tfi.extract_batch(torch_.from_numpy(imgs[list(ids)]).cuda())
    full = gallery.extract_sharded(extract, list(range(19)), 1024, torch_.device("cuda"), batch=8)
    assert tuple(full.shape) == (19, 1024)
    assert torch_.equal(full[8:16], extract(range(8, 16)))
    tfi.close_session()


def test_pca_variant_matches_sklearn_pipeline(torch_):
    """facerec_test.py:421 'k-NN+PCA': Pipeline(PCA(n), KNeighborsClassifier(1)) scored on the same split."""
    from sklearn.decomposition import PCA
    from sklearn.neighbors import KNeighborsClassifier
    from sklearn.pipeline import Pipeline
    from hse_facerec_tf_amd import identification
    X, y = oid.synthetic_gallery(80, 64, 11, 1.2)
    Xn, y2, kept = oid.filter_and_encode(X, y)
    train, test = oid.split_indices(Xn, y2)
    pipe = Pipeline(steps=[('pca', PCA(n_components=20)), ('classifier', KNeighborsClassifier(n_neighbors=1, p=2))])
    pipe.fit(Xn[train], y2[train])
    want = pipe.predict(Xn[test])
    r = identification.one_nn_identification(X, y, pca_components=20)
    assert (r["y_pred"] == want).mean() > 0.98            # PCA sign/rounding may flip a near-tie
    assert abs(r["accuracy"] - float((want == y2[test]).mean())) < 0.02


def test_config1_plumbing_100_lfw_sized_crops(torch_, tmp_path):
    """BASELINE configs[0] (the reference's own plumbing case): 100 LFW-sized (250 x 250) JPEG crops of 20 subjects through the
    file-path API -- dataset walk, labels, pipelined extract_files, npz cache, then the 1-NN protocol of facerec_test.py:401-432
    -- each image's row equal, bit for bit, to the reference-shaped per-image call."""
    from PIL import Image
    from hse_facerec_tf_amd import TensorFlowInference, extract_dataset, identification
    from conftest import MODEL_PB
    rs = np.random.RandomState(123)                                        # the reference's own seed (facerec_test.py:22)
    for s_ in range(20):
        d = tmp_path / ("subject_%02d" % s_)
        d.mkdir()
        base = rs.randint(0, 256, (25, 25, 3)).astype(np.float32)
        for i in range(5):
            im = np.kron(base, np.ones((10, 10, 1), np.float32)) + rs.randn(250, 250, 3) * 20
            Image.fromarray(np.clip(im, 0, 255).astype(np.uint8)).save(str(d / ("%d.jpg" % i)), quality=92)
    tfi = TensorFlowInference(MODEL_PB, 'input_1:0', 'global_pooling/Mean:0', input_size=(192, 192), max_batch=32)
    X, y = extract_dataset(tfi, str(tmp_path), str(tmp_path / "feats.npz"), batch=32)
    assert X.shape == (100, 1024) and list(np.bincount(y)) == [5] * 20
    for i in (0, 31, 32, 99):
        d, f = divmod(i, 5)
        one = tfi.extract_features(str(tmp_path / ("subject_%02d" % d) / ("%d.jpg" % f)))
        assert float(np.abs(one - X[i]).max()) <= 2e-5 * float(np.abs(one).max())     # bytes-in batched path vs floats-in reference path
    r = identification.one_nn_identification(X, y)
    assert len(r["test"]) == 50 and r["num_classes"] == 20 and 0.0 <= r["accuracy"] <= 1.0
    assert r["accuracy"] > 0.5                                             # blocky per-subject patterns are easy to tell apart
    tfi.close_session()


def test_extract_dataset_walk_cache_and_labels(torch_, tmp_path):
    """facerec_test.py:377-401 on a tiny synthetic 'LFW': directory walk, labels, batched extract, npz cache."""
    from PIL import Image
    from hse_facerec_tf_amd import TensorFlowInference, extract_dataset
    from conftest import MODEL_PB
    rs = np.random.RandomState(0)
    for person, k in (("carol", 2), ("alice", 3), ("bob", 1)):
        (tmp_path / person).mkdir()
        for i in range(k):
            Image.fromarray(rs.randint(0, 256, (250, 250, 3), dtype=np.uint8)).save(str(tmp_path / person / ("%d.jpg" % i)))
    tfi = TensorFlowInference(MODEL_PB, 'input_1:0', 'global_pooling/Mean:0', input_size=(96, 96), max_batch=4)
    cache = str(tmp_path / "feats.npz")
    X, y = extract_dataset(tfi, str(tmp_path), cache, batch=4)
    assert X.shape == (6, 1024) and list(y) == [0, 0, 0, 1, 2, 2]          # sorted subjects: alice, bob, carol
    one = tfi.extract_features(str(tmp_path / "alice" / "1.jpg"))
    assert float(np.abs(one - X[1]).max()) <= 2e-5 * float(np.abs(one).max())      # batched (bytes in) == per-image path (floats in) to round-off
    tfi.close_session()
    X2, y2 = extract_dataset(None, str(tmp_path), cache)                    # cache hit: extractor is not touched
    assert np.array_equal(X, X2) and np.array_equal(y, y2)


@pytest.mark.parametrize("n,m,d", [(100, 100, 1024), (65, 130, 64), (1, 1, 8), (257, 33, 2048)])
def test_pairwise_distances_vs_sklearn(torch_, n, m, d):
    from sklearn.metrics import pairwise_distances
    from hse_facerec_tf_amd import ops
    rs = np.random.RandomState(n + m)
    x = rs.randn(n, d).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = rs.randn(m, d).astype(np.float32)
    y /= np.linalg.norm(y, axis=1, keepdims=True)
    got = ops.pairwise_distances(torch_.from_numpy(x).cuda(), torch_.from_numpy(y).cuda()).cpu().numpy()
    want = pairwise_distances(x.astype(np.float64), y.astype(np.float64))
    assert np.abs(got - want).max() < 1e-5                     # distances here are ~1.4: squared-form error is ~1e-7
    if n == m:
        xs = torch_.from_numpy(x).cuda()
        self_d = ops.pairwise_distances(xs).cpu().numpy()
        assert np.all(np.diag(self_d) == 0.0) and np.abs(self_d - pairwise_distances(x.astype(np.float64))).max() < 1e-5
        assert np.abs(self_d - self_d.T).max() < 1e-5


def test_feature_distance_matrix_matches_reference_loop(torch_):
    """process_photos.py:46-60 literally (the O(N^2) Python double loop) on a small album."""
    from hse_facerec_tf_amd import identification
    rs = np.random.RandomState(3)
    feats = rs.rand(12, 1024).astype(np.float32)
    born = rs.randint(1950, 2010, 12)
    years = rs.randint(2012, 2019, 12)

    def feature_distance(i, j):
        dist = np.sqrt(np.sum((feats[i] - feats[j]) ** 2))
        max_year = max(years[i], years[j])
        cur_age_i, cur_age_j = max_year - born[i], max_year - born[j]
        age_dist = (cur_age_i - cur_age_j) ** 2 / (cur_age_i + cur_age_j)
        return [dist, age_dist * 0.1]
    pair = np.array([[feature_distance(i, j) for j in range(12)] for i in range(12)])
    want = np.clip(np.sum(pair, axis=2), a_min=0, a_max=None)
    got = identification.feature_distance_matrix(feats, born, years)
    assert np.abs(got - want).max() < 2e-4


def test_nn1_at_full_lfw_size_vs_sklearn_fixture_and_fp64(torch_):
    """BASELINE configs[4]'s identification stage at its real size -- 9164 embeddings of 1680 persons -> 4582 probes x
    4582 gallery rows x 1024-D (facerec_test.py:401-432) -- against scikit-learn's own split / predictions
    (tests/golden/nn1_lfw.npz) and against an fp64 brute force of every distance."""
    from hse_facerec_tf_amd import gallery, identification
    z = np.load(os.path.join(GOLDEN, "nn1_lfw.npz"))
    y = gallery.lfw_like_labels(int(z["n"]), int(z["n_classes"]))
    X = oid.embeddings_for_labels(y, int(z["dim"]), int(z["seed"]), float(z["noise"]))
    tm = {}
    r = identification.one_nn_identification(X, y, timings=tm)
    assert tm["nn1_shape"] == (4582, 4582, 1024) and r["num_classes"] == 1680 and len(r["indices"]) == 9164
    assert np.array_equal(r["train"], z["train"]) and np.array_equal(r["test"], z["test"])        # same split as scikit-learn
    # the same protocol with the label work started beforehand in a thread (what bench.py's config-5 leg does)
    r2 = identification.one_nn_identification(X, y, split=identification.start_split(y))
    assert all(np.array_equal(r2[k], r[k]) for k in ("indices", "y", "train", "test", "y_pred", "nn_index")) and r2["accuracy"] == r["accuracy"]
    # fp64 brute force of all 21 M distances
    Xn = X.astype(np.float64)
    Xn /= np.linalg.norm(Xn, axis=1, keepdims=True)
    d2 = ((Xn[r["test"]] ** 2).sum(1)[:, None] + (Xn[r["train"]] ** 2).sum(1)[None, :] - 2.0 * Xn[r["test"]] @ Xn[r["train"]].T)
    best = d2.min(axis=1)
    part = np.partition(d2, 1, axis=1)
    clear = (part[:, 1] - part[:, 0]) > 2e-6                 # probes whose nearest neighbour is not an fp32-level tie
    assert clear.mean() > 0.995
    assert np.array_equal(r["nn_index"][clear], d2.argmin(axis=1)[clear])
    assert np.array_equal(r["nn_index"][clear], z["nn_index"][clear])                               # == KNeighborsClassifier
    assert np.all(d2[np.arange(4582), r["nn_index"]] <= best + 2e-6)                                # a tie pick is still a nearest row
    assert np.abs(r["nn_dist"] - np.sqrt(np.maximum(best, 0))).max() < 1e-4
    assert abs(r["accuracy"] - float(z["accuracy"])) <= (~clear).sum() / 4582.0 + 1e-12
    if clear.all():
        assert r["accuracy"] == pytest.approx(float(z["accuracy"]), abs=1e-12)


def _protocol_fixture():
    z = np.load(os.path.join(GOLDEN, "protocols.npz"))
    X, y = oid.synthetic_gallery(int(z["n_classes"]), int(z["dim"]), int(z["seed"]), float(z["noise"]))
    Xn, y2, kept = oid.filter_and_encode(X, y)
    assert np.array_equal(y2, z["y"])
    return z, X[kept], Xn


def _near_tie_only(A, probe, gallery, got_idx, want_idx, tol):
    """Every probe whose nearest gallery row differs from the fixture's must be an fp tie in fp64 (distance gap <= tol)."""
    bad = np.nonzero(got_idx != want_idx)[0]
    A = A.astype(np.float64)
    for b in bad:
        d_got = ((A[probe[b]] - A[gallery[got_idx[b]]]) ** 2).sum()
        d_want = ((A[probe[b]] - A[gallery[want_idx[b]]]) ** 2).sum()
        assert abs(d_got - d_want) <= tol * max(d_want, 1.0), (b, d_got, d_want)
    return len(bad)


def test_gallery_probe_identification_matches_sklearn_fixture(torch_):
    """facerec_test.py:260-288 on the GPU vs KNeighborsClassifier(1).fit(gallery).predict(probe) frozen from scikit-learn:
    the reference's un-normalised call and the normalised variant."""
    from hse_facerec_tf_amd import identification
    z, Xraw, Xn = _protocol_fixture()
    g, p = z["gallery"], z["probe"]
    r = identification.gallery_probe_identification(Xraw[g], z["y"][g], Xraw[p], z["y"][p])
    assert _near_tie_only(Xraw, p, g, r["nn_index"], z["gp_nn_index"], 1e-6) == 0          # margin 0.35 on d2 ~ 570: no ties
    assert np.array_equal(r["y_pred"], z["gp_pred"])
    assert 100.0 * r["accuracy"] == pytest.approx(float(z["gp_accuracy_percent"]), abs=1e-9)
    assert np.abs(r["nn_dist"] - z["gp_nn_dist"]).max() < 1e-3 * float(z["gp_nn_dist"].max())
    rn = identification.gallery_probe_identification(torch_.from_numpy(Xraw[g]).cuda(), z["y"][g], torch_.from_numpy(Xraw[p]).cuda(),
                                                     z["y"][p], normalize=True)
    ties = _near_tie_only(Xn, p, g, rn["nn_index"], z["gpn_nn_index"], 2e-6)
    assert ties <= 2 and int((rn["y_pred"] != z["gpn_pred"]).sum()) <= ties
    with pytest.raises(ValueError):
        identification.gallery_probe_identification(Xraw[g], z["y"][g][:-1], Xraw[p], z["y"][p])


def test_single_image_per_class_cross_validation_matches_sklearn_fixture(torch_):
    """classifier_tester with sss = get_single_image_per_class_cv(y) (facerec_test.py:199-207, 177-197): ten splits, one
    gallery image per class, accuracies vs scikit-learn's cross_validate."""
    from hse_facerec_tf_amd import identification
    z, Xraw, Xn = _protocol_fixture()
    cv = identification.single_image_per_class_splits(z["y"], 10, 0)
    r = identification.cross_validated_1nn(Xraw, z["y"], cv)                               # normalises on the device (:401)
    assert r["accuracies"].shape == (10,)
    # a near tie (fp64 gap 7e-6 in this fixture) may flip one probe of a split: at most one probe per split may differ
    for i, (tr, te) in enumerate(cv):
        assert abs(r["accuracies"][i] - z["accuracies"][i]) <= 1.0 / len(te) + 1e-12
    assert int((np.abs(r["accuracies"] - z["accuracies"]) > 1e-12).sum()) <= 2
    assert r["mean"] == pytest.approx(float(z["accuracies"].mean()), abs=2e-4)
