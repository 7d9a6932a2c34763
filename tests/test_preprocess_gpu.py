"""GPU parity for the device-side preprocessing: bit-exact against Pillow itself and
against the oracle's restatement of facerec_test.py:80-112 / facial_analysis.py:95-107."""
import os

import numpy as np
import pytest
from PIL import Image

from oracle import pipeline as opl

from conftest import GOLDEN, MODEL_PB, TEST_IMAGE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("n,H,W,s", [(3, 250, 250, 192), (2, 250, 250, 224), (1, 588, 784, 224), (2, 100, 80, 192), (4, 192, 192, 192)])
@pytest.mark.parametrize("bgr,imagenet", [(True, True), (True, False), (False, True)])
def test_pil_path_bit_exact(torch_, n, H, W, s, bgr, imagenet):
    from hse_facerec_tf_amd import preprocess_device as pd
    imgs = np.random.RandomState(H + s).randint(0, 256, (n, H, W, 3)).astype(np.uint8)
    got = pd.preprocess_pil(imgs, (s, s), bgr, imagenet).cpu().numpy()
    want = np.stack([opl.preprocess_image(im, s, s, bgr, imagenet) for im in imgs]).astype(np.float32)   # float64 -> float32 feed
    assert got.dtype == np.float32 and np.array_equal(got, want)


@pytest.mark.parametrize("n,H,W,oh,ow", [(3, 250, 250, 192, 192), (2, 250, 250, 224, 224), (1, 588, 784, 224, 224), (2, 100, 80, 192, 192),
                                          (4, 192, 192, 192, 192), (5, 251, 249, 96, 96), (1, 33, 47, 64, 40), (2, 250, 250, 190, 190),
                                          (1, 1200, 900, 192, 192), (7, 250, 250, 100, 96)])
def test_raw_uint8_resize_is_pillows_bytes(torch_, n, H, W, oh, ow):
    """preprocess_pil(raw_u8=True) -- what Engine.forward_u8 is fed -- against PIL.Image.resize(BILINEAR) itself, byte for byte:
    the fused one-kernel form (ow % 4 == 0 and the band fits LDS), its fallbacks (ow % 4 != 0: 190; a band too large for LDS:
    1200 x 900 -> 192), enlarging, odd sizes, the last image's rows running to the very end of the buffer."""
    from hse_facerec_tf_amd import preprocess_device as pd
    imgs = np.random.RandomState(H + ow).randint(0, 256, (n, H, W, 3)).astype(np.uint8)
    got = pd.preprocess_pil(imgs, (oh, ow), raw_u8=True)
    assert got.dtype == torch_.uint8 and tuple(got.shape) == (n, oh, ow, 3)
    want = np.stack([np.asarray(Image.fromarray(im).resize((ow, oh), resample=Image.BILINEAR)) for im in imgs])
    assert np.array_equal(got.cpu().numpy(), want)
    again = pd.preprocess_pil(torch_.from_numpy(imgs).cuda(), (oh, ow), raw_u8=True)
    assert torch_.equal(got, again)


@pytest.mark.parametrize("H,W", [(37, 53), (300, 200), (224, 224), (588, 784), (64, 64)])
def test_cv_path_bit_exact(torch_, H, W):
    from hse_facerec_tf_amd import preprocess_device as pd
    imgs = np.random.RandomState(H).randint(0, 256, (2, H, W, 3)).astype(np.uint8)
    got = pd.preprocess_cv(imgs, (224, 224)).cpu().numpy()
    want = np.concatenate([opl.age_gender_preprocess(im, 224, 224) for im in imgs])
    assert np.array_equal(got, want)


def test_reference_image_through_device_preprocessing_matches_golden(torch_):
    from hse_facerec_tf_amd import FacialImageProcessing, TensorFlowInference
    z = np.load(os.path.join(GOLDEN, "e2e_test_image.npz"))
    img = opl.imread_rgb(TEST_IMAGE)
    tfi = TensorFlowInference(MODEL_PB, 'input_1:0', 'global_pooling/Mean:0', input_size=(192, 192), max_batch=4)
    f_dev = tfi.extract_images(img[None]).cpu().numpy()[0]
    f_host = tfi.extract_features(TEST_IMAGE)
    # the resized bytes are Pillow's bit for bit (tests above); the engine takes them as bytes (forward_u8) where the host path
    # feeds float32(bytes - mean): same features to fp32 round-off
    assert float(np.abs(f_dev - f_host).max()) <= 2e-5 * float(np.abs(f_host).max())
    assert np.abs(f_dev - z["feat_192"]).max() / np.abs(z["feat_192"]).max() < 1e-4
    a = tfi.extract_files([TEST_IMAGE, TEST_IMAGE], device_preprocess=True)
    b = tfi.extract_files([TEST_IMAGE, TEST_IMAGE], device_preprocess=False)
    assert float(np.abs(a - b).max()) <= 2e-5 * float(np.abs(b).max())
    tfi.close_session()
    dev = FacialImageProcessing(mtcnn_detector=False, device_preprocess=True)
    host = FacialImageProcessing(mtcnn_detector=False, device_preprocess=False)
    bgr = np.ascontiguousarray(img[..., ::-1])
    r1 = dev.process_image(bgr, bounding_boxes=z["boxes"])
    r2 = host.process_image(bgr, bounding_boxes=z["boxes"])
    f1, f2 = np.asarray(r1[4]), np.asarray(r2[4])
    assert float(np.abs(f1 - f2).max()) <= 2e-5 * float(np.abs(f2).max())
    assert np.allclose(np.asarray(r1[3]), np.asarray(r2[3]), rtol=0, atol=2e-6) and np.allclose(r1[2], r2[2], rtol=0, atol=1e-3)
    dev.close(); host.close()
