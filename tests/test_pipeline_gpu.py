"""GPU test of the files -> features pipeline (facerec_test.py:394 at scale): threaded decode, pinned
staging, double-buffered upload, device preprocessing + forward -- bit-identical to the reference-shaped serial path."""
import os
import time

import numpy as np
import pytest

from conftest import MODEL_PB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def jpegs(tmp_path_factory):
    from PIL import Image
    d = tmp_path_factory.mktemp("faces")
    rs = np.random.RandomState(7)
    paths = []
    base = rs.randint(0, 256, (8, 250, 250, 3), dtype=np.uint8)
    for i in range(300):
        im = np.roll(base[i % 8], i, axis=1)
        if i % 37 == 5:
            im = im[:200, :180]                         # a few files of another size: grouped per size inside a chunk
        p = str(d / ("%04d.jpg" % i))
        Image.fromarray(im).save(p, quality=90)
        paths.append(p)
    return paths


def test_pipelined_extract_files_is_bit_identical_to_the_serial_path(jpegs):
    import torch
    from hse_facerec_tf_amd import TensorFlowInference
    tfi = TensorFlowInference(MODEL_PB, 'input_1:0', 'global_pooling/Mean:0', input_size=(96, 96), max_batch=64)
    st = {}
    X = tfi.extract_files(jpegs, batch=64, stats=st)
    assert X.shape == (300, 1024) and st["chunks"] == 5 and st["seconds"] > 0
    # (a) the per-image path of the reference: preprocess_image on the host + one run per file (facerec_test.py:114-122)
    for i in (0, 5, 42, 63, 64, 191, 299):
        assert np.array_equal(tfi.extract_features(jpegs[i]), X[i]), i
    # (b) the serial batched path with host preprocessing
    Y = tfi.extract_files(jpegs, batch=64, device_preprocess=False)
    assert np.array_equal(X, Y)
    # (c) other chunkings / worker counts change nothing
    assert np.array_equal(tfi.extract_files(jpegs, batch=17, workers=2), X)
    assert tfi.extract_files([], batch=8).shape == (0, 1024)
    tfi.close_session()


def test_pipeline_keeps_up_with_the_decoders(jpegs):
    """Files-inclusive throughput is bounded by the host's JPEG decoders; the pipeline must stay within 2x of the aggregate
    decode rate of the same thread pool (the GPU side is two orders of magnitude faster)."""
    from concurrent.futures import ThreadPoolExecutor
    from hse_facerec_tf_amd import TensorFlowInference, preprocess
    paths = jpegs * 4
    workers = max(1, min(len(os.sched_getaffinity(0)), 32))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        list(pool.map(preprocess.imread_rgb, paths[:64]))
        t0 = time.perf_counter()
        list(pool.map(preprocess.imread_rgb, paths))
        t_dec = time.perf_counter() - t0
    tfi = TensorFlowInference(MODEL_PB, 'input_1:0', 'global_pooling/Mean:0', input_size=(192, 192), max_batch=256)
    tfi.extract_files(paths[:256], batch=256)          # warm-up
    st = {}
    tfi.extract_files(paths, batch=256, stats=st)
    tfi.close_session()
    assert st["seconds"] < 2.0 * t_dec + 0.25, (st, t_dec)
