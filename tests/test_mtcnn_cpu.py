"""CPU suite, part 8: host logic of the MTCNN cascade (box utilities, INTER_AREA resize) against the oracle's
restatement of facial_analysis.py:354-476, and the oracle cascade against its own committed fixture."""
import os

import numpy as np
import pytest

from hse_facerec_tf_amd import mtcnn as pm
from hse_facerec_tf_amd import preprocess
from oracle import mtcnn as om

from conftest import GOLDEN, TEST_IMAGE


def random_boxes(rs, n, w=640, h=480):
    x1 = rs.uniform(-30, w - 20, n)
    y1 = rs.uniform(-30, h - 20, n)
    bw = rs.uniform(12, 200, n)
    bh = rs.uniform(12, 200, n)
    return np.stack([x1, y1, x1 + bw, y1 + bh, rs.uniform(0.6, 1, n)], axis=1)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_box_utilities_match_reference_semantics(seed):
    rs = np.random.RandomState(seed)
    b = random_boxes(rs, 60)
    for thr, method, use_min in ((0.5, 'Union', False), (0.7, 'Union', False), (0.7, 'Min', True)):
        assert list(pm._iou_suppress(b, thr, use_min)) == list(om.nms(b.copy(), thr, method))
    reg = rs.randn(60, 4) * 0.1
    assert np.array_equal(pm._regress(b, reg), om.bbreg(b.copy(), reg.copy()))
    assert np.array_equal(pm._square(b), om.rerec(b.copy()))
    sq = pm._square(b)
    sq[:, 0:4] = np.fix(sq[:, 0:4])
    bw, bh, x1, y1, x2, y2, tx1, ty1, tx2, ty2 = pm._crop_windows(sq, 640, 480)
    dy, edy, dx, edx, y, ey, x, ex, tmpw, tmph = om.pad(sq.copy(), 640, 480)
    for got, want in ((bw, tmpw), (bh, tmph), (x1, x), (y1, y), (x2, ex), (y2, ey), (tx1, dx), (ty1, dy), (tx2, edx), (ty2, edy)):
        assert np.array_equal(got, want)
    assert pm._iou_suppress(np.empty((0, 9)), 0.5, False).size == 0


@pytest.mark.parametrize("n", [2, 5, 9, 16])
def test_nms_score_ties_follow_the_references_own_argsort(n):
    """bit-equal scores (softmax saturated to 1.0f).  The reference calls np.argsort(s) and picks I[-1]; on lists of
    <= 16 elements (the usual R-/O-Net list) the NumPy of the reference's time sorts with a stable insertion sort, so the
    HIGHEST index among equals goes first (NumPy >= 1.25 with SIMD sorts orders ties differently: oracle.mtcnn.nms takes the
    argsort to use).  The product's rule -- descending index among ties, for every list size -- must reproduce that."""
    old_numpy_argsort = lambda s: np.argsort(s, kind="stable")
    rs = np.random.RandomState(n)
    for trial in range(20):
        b = random_boxes(rs, n, 200, 200)
        b[:, 4] = np.float32(1.0)                                  # every score equal
        if trial % 2:
            b[rs.randint(0, n), 4] = np.float32(0.99)              # ... or all but one
        for thr, method, use_min in ((0.5, 'Union', False), (0.7, 'Min', True)):
            assert list(pm._iou_suppress(b, thr, use_min)) == list(om.nms(b.copy(), thr, method, argsort=old_numpy_argsort))


@pytest.mark.parametrize("H,W,dh,dw,dt", [(588, 784, 221, 294, np.uint8), (100, 90, 24, 24, np.float64), (30, 41, 24, 24, np.float64),
                                          (20, 20, 24, 24, np.float64), (20, 30, 24, 24, np.uint8), (96, 96, 48, 48, np.uint8),
                                          (96, 64, 32, 32, np.uint8), (60, 60, 20, 20, np.float64), (48, 48, 48, 48, np.float64)])
def test_inter_area_resize_matches_oracle_bit_exactly(H, W, dh, dw, dt):
    img = np.random.RandomState(H + W).randint(0, 256, (H, W, 3)).astype(dt)
    a, b = preprocess.resize_area(img, dw, dh), om.cv2_resize_area(img, dw, dh)
    assert a.dtype == b.dtype and np.array_equal(a, b)


def test_inter_area_properties():
    flat = np.full((37, 53, 3), 91, np.uint8)
    assert np.all(preprocess.resize_area(flat, 20, 11) == 91)                       # constants are preserved
    img = np.random.RandomState(0).randint(0, 256, (64, 64, 3)).astype(np.uint8)
    half = preprocess.resize_area(img, 32, 32).astype(int)                           # exact 2x2 box means, round half up
    want = (img.reshape(32, 2, 32, 2, 3).astype(int).sum(axis=(1, 3)) + 2) >> 2
    assert np.array_equal(half, want)
    f = img.astype(np.float64)
    assert np.allclose(preprocess.resize_area(f, 16, 16), f.reshape(16, 4, 16, 4, 3).mean(axis=(1, 3)))


def test_oracle_cascade_reproduces_its_fixture():
    z = np.load(os.path.join(GOLDEN, "mtcnn_test_image.npz"))
    from oracle.pipeline import imread_rgb
    det = om.OracleMTCNN(os.path.join(os.path.dirname(GOLDEN), "..", "models", "mtcnn.pb"), minsize=32, compute_dtype=np.float32)
    boxes, points = det.detect(imread_rgb(TEST_IMAGE))
    assert boxes.shape == (4, 5)                                                      # 4 faces, as in the reference notebook
    assert np.abs(boxes - z["boxes"]).max() < 1e-6 and np.abs(points - z["points"]).max() < 1e-4


def test_reference_notebook_outputs_as_a_sanity_anchor():
    """The reference's notebook (age_gender_identity/AgeGenderIdentityDemo.ipynb, cell 7) prints TensorFlow's age / gender
    outputs for the four faces of test_image.jpg.  They are NOT a parity pin: the notebook ran another checkpoint
    (facial_analysis.py:45 loads age_gender_tf2_224_deep-03-0.13-0.97_new.pb; the repository ships only
    age_gender_tf2_new-01-0.14-0.92_quantized.pb), so numbers differ at the 1e-1 level.  What must hold with either
    checkpoint: the same four faces in the same order, the same male/female decisions (threshold 0.6,
    facial_analysis.py:76-81), ages within a few years and the same age ranking of the adult and the older child."""
    z = np.load(os.path.join(GOLDEN, "mtcnn_test_image.npz"))
    nb_gender = np.array([0.06864629, 0.6463263, 0.47247198, 0.23946252])
    nb_age = np.array([34.56404456496239, 8.960565209388733, 2.0537627935409546, 2.684981346130371])
    ours_gender, ours_age = z["genders"][:, 0], z["ages"]
    assert np.array_equal(ours_gender >= 0.6, nb_gender >= 0.6)
    assert np.abs(ours_age - nb_age).max() < 2.5
    assert list(np.argsort(-ours_age)[:2]) == list(np.argsort(-nb_age)[:2])       # adult, then the older child
