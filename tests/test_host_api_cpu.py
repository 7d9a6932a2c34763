"""CPU suite, part 4: the host-side mirror of the reference API (everything that does not need
the device), checked against the oracle's restatement of the same reference lines."""
import os

import numpy as np
import pytest

from hse_facerec_tf_amd import facial_analysis, identification, preprocess, tf_inference
from oracle import identification as oid
from oracle import pipeline as opl

from conftest import GOLDEN, MODEL_PB, TEST_IMAGE


def test_imread_imresize_and_model_input_match_reference_restatement():
    img = preprocess.imread_rgb(TEST_IMAGE)
    assert img.shape == (588, 784, 3) and img.dtype == np.uint8
    for bgr, imagenet in ((True, True), (True, False), (False, True)):
        ref = opl.preprocess_image(img, 192, 192, bgr, imagenet)
        got = preprocess.to_model_input(preprocess.imresize_bilinear(img, (192, 192)), bgr, imagenet)
        assert got.dtype == np.float64 and np.array_equal(got, ref)
    ref = opl.preprocess_image(img, 224, 224, True, True, crop_center=True)
    got = preprocess.to_model_input(preprocess.imresize_bilinear(preprocess.center_crop_250_128(img), (224, 224)), True, True)
    assert np.array_equal(got, ref)


def test_linear_resize_matches_oracle_restatement_bit_exactly():
    rs = np.random.RandomState(0)
    for (h, w, oh, ow) in [(37, 53, 224, 224), (300, 200, 224, 224), (224, 224, 224, 224), (500, 448, 224, 224), (5, 7, 3, 2)]:
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        assert np.array_equal(preprocess.resize_linear_u8(img, ow, oh), opl.cv2_resize_linear(img, ow, oh))


def test_constructor_error_behaviour_before_any_device_work(tmp_path):
    with pytest.raises(FileNotFoundError):
        tf_inference.TensorFlowInference(str(tmp_path / "nope.pb"), "input_1:0", "global_pooling/Mean:0")
    with pytest.raises(KeyError):       # graph.get_tensor_by_name on an unknown tensor (facerec_test.py:62)
        tf_inference.TensorFlowInference(MODEL_PB, "input_1:0", "reshape_1/Reshape:0")
    with pytest.raises(KeyError):
        tf_inference.TensorFlowInference(MODEL_PB, "input:0", "global_pooling/Mean:0")
    with pytest.raises(KeyError):
        tf_inference.TensorFlowInference(MODEL_PB, "input_1:0", "global_pooling/Mean:0",
                                         learning_phase_tensor="conv1_bn/keras_learning_phase:0")


def test_no_cpu_fallback_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tf_inference.TensorFlowInference(MODEL_PB, "input_1:0", "global_pooling/Mean:0")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        facial_analysis.FacialImageProcessing()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        identification.one_nn_identification(np.zeros((4, 8), np.float32), np.array([0, 0, 1, 1]))


def test_product_never_imports_the_oracle():
    import ast
    root = os.path.join(os.path.dirname(GOLDEN), "..", "hse_facerec_tf_amd")
    for fn in os.listdir(root):
        if fn.endswith(".py"):
            tree = ast.parse(open(os.path.join(root, fn)).read())
            for node in ast.walk(tree):
                mods = []
                if isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    mods = [node.module or ""]
                assert not any(m.split(".")[0] == "oracle" for m in mods), "%s imports the oracle" % fn


def test_age_decode_and_box_geometry():
    p = np.random.RandomState(1).dirichlet(np.ones(100)).astype(np.float32)
    assert facial_analysis.decode_age(p)[0] == opl.decode_age(p)[0]
    boxes = facial_analysis.FacialImageProcessing.face_boxes(
        [[100.7, 60.2, 260.9, 260.1, 0.99], [-5, 400, 150, 600], [50, 50, 50, 80], [770, 570, 800, 600]], 784, 588)
    assert boxes == [[90, 50, 270, 270], [0, 390, 160, 588], [760, 560, 784, 588]]    # pad 10, clip, drop empty
    assert facial_analysis.FacialImageProcessing.is_male(np.array([0.6]))[0]


def test_filter_classes_and_split_match_sklearn_protocol():
    X, y = oid.synthetic_gallery(n_classes=60, dim=32, seed=5, noise=1.0)
    Xn, y_ref, kept_ref = oid.filter_and_encode(X, y)
    kept, y_enc = identification.filter_classes(y)
    assert np.array_equal(kept, kept_ref) and np.array_equal(y_enc, y_ref)
    tr_ref, te_ref = oid.split_indices(Xn, y_ref)
    tr, te = identification.stratified_half_split(y_enc)
    assert np.array_equal(tr, tr_ref) and np.array_equal(te, te_ref)
    # the reference's O(N^2) list.count filter (facerec_test.py:408-409), literally
    y_l = list(y)
    assert [i for i, el in enumerate(y_l) if y_l.count(el) > 1] == list(kept)


def test_get_files_walks_like_the_reference(tmp_path):
    for d, files in (("bob", ["b.jpg", "a.PNG", ".hidden.jpg", "notes.txt"]), ("alice", ["1.jpeg"])):
        os.makedirs(tmp_path / d)
        for f in files:
            (tmp_path / d / f).write_bytes(b"x")
    got = tf_inference.get_files(str(tmp_path))
    assert got == [["alice", os.path.join("alice", "1.jpeg")], ["bob", os.path.join("bob", "a.PNG")],
                   ["bob", os.path.join("bob", "b.jpg")]]


def test_float32_face_preprocessing_matches_reference_arithmetic():
    """facial_analysis.py:101-107 subtracts Python floats from a float32 image (float32 arithmetic)."""
    img = np.random.RandomState(0).randint(0, 256, (300, 200, 3)).astype(np.uint8)
    got = preprocess.to_model_input(preprocess.resize_linear_u8(img, 224, 224), True, True, dtype=np.float32)
    want = opl.age_gender_preprocess(img, 224, 224)[0]
    assert got.dtype == np.float32 and np.array_equal(got, want)


def test_split_job_gives_the_serial_split_and_reraises():
    """identification.start_split: filter + StratifiedShuffleSplit in a thread (bench.py starts it under the extraction) --
    the same four arrays as the serial calls; an exception inside the thread surfaces in result()."""
    from hse_facerec_tf_amd import gallery, identification
    y = gallery.lfw_like_labels(600, 110)
    indices, y_enc = identification.filter_classes(y)
    tr, te = identification.stratified_half_split(y_enc)
    got = identification.start_split(y).result()
    assert all(np.array_equal(a, b) for a, b in zip(got, (indices, y_enc, tr, te)))
    with pytest.raises(ValueError):
        identification.start_split(np.arange(10)).result()      # no class with two samples: scikit-learn refuses the split


def test_single_image_per_class_splits_equal_numpy_seeded_reference_loop():
    """facerec_test.py:177-197: bit-equal to the fixture frozen from NumPy's GLOBAL seeded generator (the reference's own
    calls, restated in oracle/identification.py), without touching the caller's global state."""
    z = np.load(os.path.join(GOLDEN, "protocols.npz"))
    np.random.seed(77)
    before = np.random.get_state()[1].copy()
    cv = identification.single_image_per_class_splits(z["y"], n_splits=10, random_state=0)
    assert np.array_equal(np.random.get_state()[1], before)
    assert len(cv) == 10
    for i, (tr, te) in enumerate(cv):
        assert np.array_equal(tr, z["train_%d" % i]) and np.array_equal(te, z["test_%d" % i])
        assert len(tr) == len(np.unique(z["y"])) and len(tr) + len(te) == len(z["y"])
        assert np.array_equal(np.sort(z["y"][tr]), np.unique(z["y"]))          # exactly one gallery image per class
    live = oid.single_image_per_class_cv(z["y"], 10, 0)                         # and the restatement run live agrees too
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(cv, live))
    other = identification.single_image_per_class_splits(z["y"], n_splits=2, random_state=1)
    assert not np.array_equal(other[0][0], cv[0][0])


def test_load_graph_prefix_renames_like_import_graph_def():
    """facerec_test.py:41-48 / facial_analysis.py:328-332: tf.import_graph_def(graph_def, name=prefix)."""
    from hse_facerec_tf_amd import graphdef, lowering
    g0 = tf_inference.load_graph(MODEL_PB)
    g = tf_inference.load_graph(MODEL_PB, prefix="age")
    assert len(g.nodes) == len(g0.nodes)
    assert all(n.name == "age/" + m.name and n.op == m.op for n, m in zip(g.nodes, g0.nodes))
    for n, m in zip(g.nodes, g0.nodes):
        assert [r.lstrip("^") for r in n.inputs] == ["age/" + r.lstrip("^") for r in m.inputs]
        assert [r.startswith("^") for r in n.inputs] == [r.startswith("^") for r in m.inputs]
    g.get_tensor_by_name("age/input_1:0")
    with pytest.raises(KeyError):
        g.get_tensor_by_name("input_1:0")
    assert tf_inference.load_graph(MODEL_PB, prefix="age/").node("age/input_1").op == "Placeholder"
    # the prefixed graph lowers to the SAME plan bytes (names are not part of a plan)
    a = lowering.lower_graph(g0, "input_1:0", {0: "global_pooling/Mean:0"}, (96, 96)).serialize()
    b = lowering.lower_graph(g, "age/input_1:0", {0: "age/global_pooling/Mean:0"}, (96, 96)).serialize()
    assert a == b
    # several files behind one session (facial_analysis.py:55-58): one graph, disjoint prefixes; a clash raises
    full = graphdef.Graph.merged([g, tf_inference.load_graph(MODEL_PB, prefix="gender")])
    assert len(full.nodes) == 2 * len(g0.nodes)
    full.get_tensor_by_name("gender/gender_pred/Sigmoid:0")
    with pytest.raises(ValueError):
        graphdef.Graph.merged([g0, g0])


class _StubExtractor:
    """extract_files of a TensorFlowInference without a device: the feature of a file is its size and its first byte."""
    feature_dim = 2

    def __init__(self):
        self.calls = []

    def extract_files(self, paths, batch=256, crop_center=False):
        self.calls.append(list(paths))
        return np.array([[os.path.getsize(p), open(p, "rb").read(1)[0]] for p in paths], dtype=np.float32).reshape(-1, 2)


def test_extract_gallery_probe_labels_cache_and_unseen_subject(tmp_path):
    """facerec_test.py:220-258: encoder fitted on the Gallery tree, applied to the Probe tree; x_train / y_train / x_test / y_test
    cache; a probe subject the gallery lacks raises as LabelEncoder.transform does."""
    for tree, subjects in (("Gallery", {"carol": 2, "alice": 1, "bob": 3}), ("Probe", {"bob": 2, "alice": 2})):
        for s, k in subjects.items():
            os.makedirs(tmp_path / tree / s)
            for i in range(k):
                (tmp_path / tree / s / ("%d.jpg" % i)).write_bytes(bytes([65 + i]) * (1 + i + len(s)))
    stub = _StubExtractor()
    cache = str(tmp_path / "feats.npz")
    Xtr, ytr, Xte, yte = tf_inference.extract_gallery_probe(stub, str(tmp_path / "Gallery"), str(tmp_path / "Probe"), cache)
    assert list(ytr) == [0, 1, 1, 1, 2, 2] and list(yte) == [0, 0, 1, 1]          # alice 0, bob 1, carol 2 (sorted walk)
    assert Xtr.shape == (6, 2) and Xte.shape == (4, 2) and len(stub.calls) == 2
    assert stub.calls[1][0].endswith(os.path.join("Probe", "alice", "0.jpg"))
    z = np.load(cache)
    assert sorted(z.files) == ["x_test", "x_train", "y_test", "y_train"]          # facerec_test.py:258
    again = tf_inference.extract_gallery_probe(None, "/nonexistent", "/nonexistent", cache)     # :227 the cache short-circuits
    assert all(np.array_equal(a, b) for a, b in zip(again, (Xtr, ytr, Xte, yte)))
    os.makedirs(tmp_path / "Probe" / "dave")
    (tmp_path / "Probe" / "dave" / "0.png").write_bytes(b"zz")
    with pytest.raises(ValueError, match="unseen"):
        tf_inference.extract_gallery_probe(stub, str(tmp_path / "Gallery"), str(tmp_path / "Probe"))


def test_extract_dataset_subjects_file_variant(tmp_path):
    """facerec_test.py:378-380: the LFW-and-YTF protocol walks only the subjects of lfw_ytf_classes.txt, in that order."""
    for s in ("zed", "amy", "kim"):
        os.makedirs(tmp_path / "db" / s)
        (tmp_path / "db" / s / "a.jpg").write_bytes(s.encode())
        (tmp_path / "db" / s / "b.txt").write_bytes(b"no")
    (tmp_path / "classes.txt").write_text("zed\namy\n")
    assert tf_inference.get_files_of_subjects(str(tmp_path / "db"), str(tmp_path / "classes.txt")) == \
        [["zed", os.path.join("zed", "a.jpg")], ["amy", os.path.join("amy", "a.jpg")]]
    X, y = tf_inference.extract_dataset(_StubExtractor(), str(tmp_path / "db"), subjects_file=str(tmp_path / "classes.txt"))
    assert list(y) == [1, 0] and X.shape == (2, 2)                                # LabelEncoder order: amy 0, zed 1
