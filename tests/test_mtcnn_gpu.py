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


def test_one_detector_serves_two_threads(det):
    """the uploaded frame is an argument of the stage methods, not detector state, and the work tensors of the
    device box logic are per thread -- two threads (each on its own stream, as a threaded pipeline would run them) detecting
    DIFFERENT frames through one detector get what a serial run gets."""
    import threading
    import torch
    img = opl.imread_rgb(TEST_IMAGE)
    frames = [img, np.ascontiguousarray(img[:, ::-1]), np.ascontiguousarray(img[40:500, 100:700]), np.ascontiguousarray(img[::-1])]
    serial = [det(f) for f in frames]
    out = [None] * len(frames)
    errors = []

    def work(ids):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(3):
                    for i in ids:
                        out[i] = det(frames[i])
        except Exception as e:           # surfaced below: an exception in a thread must fail the test
            errors.append(e)
    ts = [threading.Thread(target=work, args=(ids,)) for ids in ((0, 2), (1, 3))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for (b0, p0), (b1, p1) in zip(serial, out):
        assert np.array_equal(b0, b1) and np.array_equal(p0, p1)


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


def test_device_pyramid_levels_are_the_host_restatement(det):
    """hsefr_mtcnn_pyramid_level against preprocess.resize_area (the INTER_AREA restatement the oracle uses) on every level of the
    reference image's pyramid and on the special cases: exact 2x2 and 3x3 box factors, enlarging, identity.  Equal value for
    value, except that a float32 box sum landing within an ulp of xx.5 may round to the other uint8 level (2 of 194 922 values
    on the first level): at most one value in 10 000, never by more than one level."""
    import torch
    from hse_facerec_tf_amd import preprocess
    img = opl.imread_rgb(TEST_IMAGE)
    h, w = img.shape[:2]
    frame = torch.from_numpy(np.ascontiguousarray(img)).to(det.device)
    sizes = [(int(np.ceil(h * s)), int(np.ceil(w * s))) for s in det.pyramid_scales(h, w)]
    sizes += [(h // 2, w // 2), (h // 3, w // 3 + (1 if w % 3 else 0)), (h, w), (h + 40, w + 25), (h // 2, w + 10)]
    for hs, ws in sizes:
        got = det._level_device(frame, h, w, hs, ws)[0].cpu().numpy()
        want = np.transpose((preprocess.resize_area(img, ws, hs) - 127.5) * 0.0078125, (1, 0, 2)).astype(np.float32)
        assert got.shape == want.shape == (ws, hs, 3)
        diff = np.abs(got - want)
        assert diff.max() <= 0.0078125 + 1e-7 and (diff > 0).mean() <= 1e-4, (hs, ws, diff.max(), int((diff > 0).sum()))


def test_device_crops_match_the_host_restatement(det):
    """hsefr_mtcnn_crops against the reference's pad() + cv2.resize(INTER_AREA) on boxes inside, across and outside the frame,
    shrinking, enlarging and at size."""
    import torch
    img = opl.imread_rgb(TEST_IMAGE)
    h, w = img.shape[:2]
    rs = np.random.RandomState(3)
    boxes = []
    for _ in range(40):
        side = int(rs.choice([10, 24, 25, 47, 48, 49, 96, 130, 300]))
        x1 = int(rs.randint(-side // 2, w - side // 2))
        y1 = int(rs.randint(-side // 2, h - side // 2))
        boxes.append([x1, y1, x1 + side - 1, y1 + side - 1 + int(rs.randint(0, 2)), 0.9])
    boxes = np.asarray(boxes, np.float64)
    for size in (24, 48):
        det.device_resize = True
        frame = torch.from_numpy(np.ascontiguousarray(img)).to(det.device)
        got = det._crops(img, boxes, size, frame).cpu().numpy()
        det.device_resize = False
        want = det._crops(img, boxes, size).cpu().numpy()
        det.device_resize = True
        assert got.shape == want.shape == (40, size, size, 3)
        assert np.abs(got - want).max() < 2e-6              # float64 box sums in another order: round-off of the final float32


def test_device_and_host_resizing_find_the_same_faces(z):
    from hse_facerec_tf_amd.mtcnn import MTCNNDetector
    img = opl.imread_rgb(TEST_IMAGE)
    a = MTCNNDetector(minsize=32, device_resize=True)(img)
    b = MTCNNDetector(minsize=32, device_resize=False)(img)
    assert a[0].shape == b[0].shape == (4, 5)
    assert np.abs(a[0] - b[0]).max() < 1e-3 and np.abs(a[1] - b[1]).max() < 1e-3
    with pytest.raises(ValueError):
        MTCNNDetector(minsize=32)(img.astype(np.float32))


@pytest.mark.parametrize("n,thr,use_min,seed", [(1, 0.5, False, 0), (37, 0.5, False, 1), (400, 0.7, False, 2), (2048, 0.7, True, 3), (300, 0.3, True, 4)])
def test_device_nms_is_the_host_nms(n, thr, use_min, seed):
    """csrc/mtcnn_post.hip's greedy NMS against mtcnn._iou_suppress: same kept indices in the same order, including score TIES
    (descending score, ascending index) and heavily overlapping clusters."""
    import torch
    from hse_facerec_tf_amd import _lib, mtcnn
    rs = np.random.RandomState(seed)
    cx, cy = rs.uniform(0, 300, n), rs.uniform(0, 300, n)
    if n > 30:            # clusters of near-duplicates
        cx[n // 2:] = cx[:n - n // 2] + rs.uniform(-3, 3, n - n // 2)
        cy[n // 2:] = cy[:n - n // 2] + rs.uniform(-3, 3, n - n // 2)
    s = rs.uniform(8, 60, n)
    score = rs.uniform(0.6, 1.0, n).astype(np.float32)
    if n > 10:
        score[::7] = score[3]                                  # exact ties
    boxes = np.stack([np.fix(cx - s), np.fix(cy - s), np.fix(cx + s), np.fix(cy + s), score.astype(np.float64)], axis=1)
    want = mtcnn._iou_suppress(boxes, thr, use_min)
    d = torch.from_numpy(boxes).cuda()
    keep = torch.empty(n, dtype=torch.int32, device="cuda")
    nk = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().hsefr_mtcnn_nms(d.data_ptr(), n, float(thr), int(use_min), keep.data_ptr(), nk.data_ptr(), _lib.current_stream_ptr()))
    got = keep[:int(nk.item())].cpu().numpy()
    assert np.array_equal(got, want)
    assert 0 < len(got) <= n


def _variants(img):
    yield "original", img
    yield "flipped", np.ascontiguousarray(img[:, ::-1])
    yield "cropped", np.ascontiguousarray(img[40:520, 100:700])
    yield "half", np.ascontiguousarray(img[::2, ::2])
    yield "dark", (img.astype(np.float32) * 0.6).astype(np.uint8)
    yield "no face", np.ascontiguousarray(img[:90, :200])


def test_device_box_logic_is_the_host_box_logic():
    """The whole cascade with candidate generation / NMS / regression / squaring / crop windows / landmarks on the GPU against the
    same cascade with that logic in NumPy (same nets, same device pyramid and crops): identical boxes and landmarks, frame by
    frame, on the reference's photo and five variants of it (one without any face)."""
    from hse_facerec_tf_amd import preprocess
    from hse_facerec_tf_amd.mtcnn import MTCNNDetector
    img = preprocess.imread_rgb(TEST_IMAGE)
    dev = MTCNNDetector(minsize=32)                     # default: device_resize and device_boxes
    host = MTCNNDetector(minsize=32, device_boxes=False)
    assert dev.device_boxes and not host.device_boxes
    total = 0
    for name, frame in _variants(img):
        bd, pd = dev(frame)
        bh, ph = host(frame)
        assert bd.shape[0] == bh.shape[0], name
        total += bd.shape[0]
        if bd.shape[0]:
            assert bd.shape == bh.shape and pd.shape == ph.shape == (10, bd.shape[0]), name
            assert np.array_equal(bd, bh), (name, np.abs(bd - bh).max())
            assert np.array_equal(pd, ph), (name, np.abs(pd - ph).max())
    assert total >= 8 and dev.host_fallbacks == 0
    with pytest.raises(ValueError):
        MTCNNDetector(minsize=32, device_resize=False, device_boxes=True)
