"""GPU parity for the MTCNN cascade vs the oracle restatement frozen in tests/golden/mtcnn_test_image.npz,
and the full drop-in ``process_image(frame)`` of facial_analysis.py:225-294 (detection + age/gender/identity)."""
import os

import numpy as np
import pytest

from oracle import pipeline as opl

from conftest import GOLDEN, TEST_IMAGE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det():
    import torch
    assert torch.cuda.is_available()
    from hse_facerec_tf_amd.mtcnn import MTCNNDetector
    return MTCNNDetector(minsize=32)


@pytest.fixture(scope="module")
def z():
    return np.load(os.path.join(GOLDEN, "mtcnn_test_image.npz"))


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


def test_three_nets_match_the_graph_oracle(det, z):
    import torch
    for name, n_out in (("pnet", 2), ("rnet", 2), ("onet", 3)):
        x = torch.from_numpy(z[name + "_x"]).cuda()
        outs = getattr(det, name)(x)
        assert len(outs) == n_out
        for i, o in enumerate(outs):
            want = z["%s_out%d" % (name, i)]
            assert tuple(o.shape) == want.shape
            assert rel(o.cpu().numpy(), want) < 1e-5, (name, i)


def test_pyramid_scales_follow_the_reference(det):
    s = det.pyramid_scales(588, 784)                      # facial_analysis.py:489-499, minsize 32
    assert len(s) == 9 and abs(s[0] - 0.375) < 1e-12 and abs(s[1] / s[0] - 0.709) < 1e-12
    assert det.pyramid_scales(20, 500) == []              # smaller than minsize: no level, no faces


def test_detection_on_the_reference_image(det, z):
    img = opl.imread_rgb(TEST_IMAGE)
    boxes, points = det(img)
    assert boxes.shape == z["boxes"].shape == (4, 5)      # the notebook shows 4 faces too (AgeGenderIdentityDemo.ipynb:109-125)
    assert points.shape == z["points"].shape == (10, 4)
    assert np.abs(boxes[:, :4] - z["boxes"][:, :4]).max() < 0.05           # pixels
    assert np.abs(boxes[:, 4] - z["boxes"][:, 4]).max() < 1e-4
    assert np.abs(points - z["points"]).max() < 0.05
    empty_boxes, empty_pts = det(np.zeros((64, 64, 3), np.uint8))
    assert empty_boxes.shape[0] == 0


def test_process_image_full_dropin(z):
    """FacialImageProcessing(mtcnn_detector=True).process_image(bgr frame) as process_photos.py:33 calls it."""
    from hse_facerec_tf_amd import FacialImageProcessing
    fip = FacialImageProcessing(print_stat=False, mtcnn_detector=True, minsize=32)
    img = opl.imread_rgb(TEST_IMAGE)
    bboxes, points, ages, genders, feats = fip.process_image(np.ascontiguousarray(img[..., ::-1]))
    assert len(bboxes) == 4 and np.asarray(points).shape == (10, 4)
    want_boxes = [[max(int(b[0]) - 10, 0), max(int(b[1]) - 10, 0), int(b[2]) + 10, int(b[3]) + 10] for b in z["boxes"]]
    assert bboxes == want_boxes
    assert np.abs(np.asarray(ages) - z["ages"]).max() < 1e-2
    assert rel(np.asarray(genders), z["genders"]) < 1e-4 and rel(np.asarray(feats), z["feats"]) < 1e-4
    assert [bool(FacialImageProcessing.is_male(g)[0]) for g in genders] == [bool(g >= 0.6) for g in z["genders"].ravel()]
    fip.close()
