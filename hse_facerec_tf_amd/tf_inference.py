"""Drop-in for the reference's ``TensorFlowInference`` (facerec_test.py:50-125) and its model
registry ``get_tf_face_recognizer`` (facerec_test.py:209-218), running on libhsefr instead of
``tf.Session``.  Same constructor signature, attributes (``w``, ``h``), methods and error
behaviour; ``facial_clustering_test.py:16,291`` imports and constructs it by these names.

Added for the metric (the reference only has the batch-1 loop, facerec_test.py:394):
``extract_batch`` (device tensor in, device tensor out) and ``extract_files``.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, preprocess
from .engine import Engine
from .graphdef import Graph, read_graph
from .lowering import OUT_AGE, OUT_FEATURES, OUT_GENDER, Plan, lower_graph

img_extensions = ['.jpg', '.jpeg', '.png']   # facerec_test.py:33


def is_image(path):                          # facerec_test.py:34-36
    _, file_extension = os.path.splitext(path)
    return file_extension.lower() in img_extensions


def get_files(db_dir):                       # facerec_test.py:38-39 (os.walk order made deterministic)
    return [[d, os.path.join(d, f)] for d in sorted(next(os.walk(db_dir))[1])
            for f in sorted(next(os.walk(os.path.join(db_dir, d)))[2]) if not f.startswith(".") and is_image(f)]


def load_graph(frozen_graph_filename, prefix='') -> Graph:   # facerec_test.py:41-48, facial_analysis.py:328-332
    """``tf.import_graph_def(graph_def, name=prefix)``: with a prefix every node is addressed as ``prefix/<name>``
    (the reference imports its separate age and gender files as 'age/...' and 'gender/...': facial_analysis.py:48-58)."""
    return read_graph(frozen_graph_filename).with_prefix(prefix)


class TensorFlowInference:
    def __init__(self, frozen_graph_filename, input_tensor, output_tensor, learning_phase_tensor=None,
                 convert2BGR=True, imageNetUtilsMean=True, additional_input_value=0,
                 input_size: Optional[Tuple[int, int]] = None, max_batch: int = 256, device: Optional[int] = None,
                 dtype: str = "auto", input_bound: Optional[float] = 256.0, latency_plan: bool = False):
        if str(frozen_graph_filename).lower().endswith((".h5", ".hdf5")):
            # Keras weights of MobileNet-v1 (models/vgg2_mobilenet.h5, facerec_test.py:322-334: model.load_weights + the
            # 'reshape_1' output) read without an HDF5 library and turned into the frozen graph of the same model
            from .h5weights import keras_mobilenet_graph
            graph = keras_mobilenet_graph(frozen_graph_filename, int((input_size or (192, 192))[0]))
        else:
            graph = load_graph(frozen_graph_filename, '')
        self.graph = graph
        # graph.get_tensor_by_name semantics: KeyError for unknown names (facerec_test.py:60-64)
        in_node, _ = graph.get_tensor_by_name(input_tensor)
        graph.get_tensor_by_name(output_tensor)
        feeds = {}
        if learning_phase_tensor:
            graph.get_tensor_by_name(learning_phase_tensor)
            feeds[learning_phase_tensor] = additional_input_value   # fed on every run at facerec_test.py:118-119
        shape = graph.placeholder_shape(in_node.name)
        if shape is None or len(shape) != 4:           # facerec_test.py:66-70
            w = h = 160
        else:
            _, w, h, _ = shape
        if input_size is not None:                     # extension: the trunk is fully convolutional
            w, h = input_size
        self.w, self.h = int(w), int(h)
        if self.w <= 0 or self.h <= 0:
            raise ValueError("input placeholder has no static size; pass input_size=(w, h)")
        self.convert2BGR = convert2BGR
        self.imageNetUtilsMean = imageNetUtilsMean
        self.additional_input_value = additional_input_value
        self.tf_input_image, self.tf_output_features = input_tensor, output_tensor
        self.tf_learning_phase = learning_phase_tensor
        # NB the reference unpacks the NHWC placeholder shape as (_, w, h, _) and feeds [h?, w?]:
        # rows = self.w.  All models in scope are square.
        # dtype: 'f32' = fp32-grade results: the MobileNet kernels when the graph is MobileNet-shaped, else the general
        # exact-fp32 kernels (ResNet-style graphs at the 1e-4 bar, an order of magnitude slower than bf16);
        # 'bf16' = bf16-MFMA kernels (ResNet-style graphs, BASELINE config 3), 'auto' = the MobileNet fp32 kernels when
        # they cover the graph, else bf16.  self.dtype names what was chosen ('f32', 'f32g' = general fp32, 'bf16').
        # input_bound: what preprocess_image guarantees about the values this class feeds -- uint8 pixels minus a BGR mean lie
        # in [-131.1, 151.1], the non-BGR branch in [-1, 1] (facerec_test.py:93-110): |x| < 256.  The fused stem uses it
        # (csrc/stem3_fused.hip) and CHECKS it on the device.  extract_batch with values outside it raises: at once for NumPy
        # input; for a CUDA tensor (asynchronous) at the next call into this object after that forward has finished, at
        # check_input_bound() or at close_session() -- never silently.  None = no assumption about extract_batch's input
        # (exact-fp32 first convolution).
        self.input_bound = input_bound
        self._ovf_host = None              # pinned int32 ring the device-side bound flag is copied into (CUDA-tensor callers)
        self._ovf_pending = []             # [(event, slot)] in stream order
        self._ovf_next = 0
        from .lowering import LoweringError
        # uint8 entry of the engine (Engine.forward_u8): the float conversion, channel reversal and mean of :95-106 folded into
        # the first kernel -- extract_images / extract_files then hand the resized bytes over as they are
        u8_mean = None
        if convert2BGR and input_bound is not None and input_bound >= 256.0:
            u8_mean = tuple(float(m) for m in (preprocess.IMAGENET_CAFFE_BGR_MEAN if imageNetUtilsMean else preprocess.VGGFACE2_BGR_MEAN))
        if dtype == "auto":
            try:
                self.plan: Plan = lower_graph(graph, input_tensor, {OUT_FEATURES: output_tensor}, (self.w, self.h), feeds,
                                              input_bound=input_bound, u8_mean_bgr=u8_mean)
                dtype = "f32"
            except LoweringError:
                self.plan = lower_graph(graph, input_tensor, {OUT_FEATURES: output_tensor}, (self.w, self.h), feeds, dtype="bf16")
                dtype = "bf16"
        elif dtype == "f32":
            try:
                self.plan = lower_graph(graph, input_tensor, {OUT_FEATURES: output_tensor}, (self.w, self.h), feeds,
                                        input_bound=input_bound, u8_mean_bgr=u8_mean)
            except LoweringError:
                self.plan = lower_graph(graph, input_tensor, {OUT_FEATURES: output_tensor}, (self.w, self.h), feeds, dtype="f32g")
                dtype = "f32g"
        else:
            self.plan = lower_graph(graph, input_tensor, {OUT_FEATURES: output_tensor}, (self.w, self.h), feeds, dtype=dtype)
        self.dtype = dtype
        small_plan = None
        self.latency_plan = bool(latency_plan) and dtype == "f32"
        if self.latency_plan:
            # latency_plan=True: extract_features -- the reference's one-image-per-run call (facerec_test.py:114-122) -- runs a
            # second lowering of the same graph whose kernels do not need hundreds of images to fill the chip (Engine.__init__:
            # 0.17 instead of 0.26 ms per image on the device).  OFF by default: its results differ from the bulk paths' in the
            # last bits (3e-7), and by default extract_files(paths) IS [extract_features(p) for p in paths], bit for bit.
            small_plan = lower_graph(graph, input_tensor, {OUT_FEATURES: output_tensor}, (self.w, self.h), feeds,
                                     input_bound=input_bound, u8_mean_bgr=u8_mean, presplit="none")
        self.engine = Engine(self.plan, max_batch=max_batch, device=device, small_plan=small_plan)
        self.tf_sess = self.engine           # attribute name kept for callers that poke at it
        self.feature_dim = self.engine.out_elems[OUT_FEATURES]

    # ---- facerec_test.py:80-112 ---------------------------------------------------------------
    def preprocess_image(self, img_filepath, crop_center):
        img = preprocess.imread_rgb(img_filepath)
        if crop_center:
            img = preprocess.center_crop_250_128(img)
        x = preprocess.imresize_bilinear(img, (self.w, self.h))
        return preprocess.to_model_input(x, self.convert2BGR, self.imageNetUtilsMean, dtype=float)

    # ---- facerec_test.py:114-122 ----------------------------------------------------------------
    def extract_features(self, img_filepath, crop_center=False):
        torch = _lib.require_gpu()
        if self.engine.accepts_u8:
            # bytes in: decode (+ the optional centre crop) on the host as the reference does, then misc.imresize on the device
            # (bit-exact with Pillow: tests/test_preprocess_gpu.py) and float conversion / BGR / mean inside the first kernel --
            # no float64 image on the host, a 4x smaller upload; same features to 2e-6 (tests/test_stem4_gpu.py)
            img = preprocess.imread_rgb(img_filepath)
            if crop_center:
                img = preprocess.center_crop_250_128(img)
            return self.extract_images(img[None], latency=self.latency_plan).cpu().numpy().reshape(-1)
        x = self.preprocess_image(img_filepath, crop_center)
        x = np.expand_dims(x, axis=0)
        xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self.engine.device)
        preds = self.engine.forward(xd, (OUT_FEATURES,), latency=self.latency_plan)["features"]
        return preds.cpu().numpy().reshape(-1)

    def _check_bound(self, x_np) -> None:
        if self.input_bound is not None and x_np.size and not float(np.abs(x_np).max()) < self.input_bound:
            raise ValueError("input values reach %g: outside the bound %g this extractor was built for (preprocess_image's output "
                             "is; pass input_bound=None or a larger bound for other data)" % (float(np.abs(x_np).max()), self.input_bound))

    def _bound_check_queue(self) -> None:
        """After a forward on caller-supplied CUDA values: queue an asynchronous read-and-clear of the device's bound flag."""
        if self.input_bound is None:
            return
        torch = _lib.require_gpu()
        if self._ovf_host is None:
            self._ovf_host = torch.zeros(32, dtype=torch.int32).pin_memory()
        if len(self._ovf_pending) >= self._ovf_host.numel():
            self._bound_check_poll(wait_oldest=True)
        slot = self._ovf_next
        self._ovf_next = (slot + 1) % self._ovf_host.numel()
        self.engine.input_overflow_async(self._ovf_host.data_ptr() + 4 * slot)
        with torch.cuda.device(self.engine.device):
            ev = torch.cuda.Event()
            ev.record()
        self._ovf_pending.append((ev, slot))

    def _bound_check_poll(self, wait: bool = False, wait_oldest: bool = False) -> None:
        """Raise for any finished forward that fed values outside input_bound (the features it returned are meaningless)."""
        bad = False
        while self._ovf_pending:
            ev, slot = self._ovf_pending[0]
            if wait or wait_oldest:
                ev.synchronize()
                wait_oldest = False
            elif not ev.query():
                break
            self._ovf_pending.pop(0)
            if int(self._ovf_host[slot]):
                self._ovf_host[slot] = 0
                bad = True
        if bad:
            raise ValueError("an earlier extract_batch on a CUDA tensor fed values outside the bound %g this extractor was built "
                             "for: the features it returned are meaningless (pass input_bound=None or a larger bound for such data)"
                             % self.input_bound)

    def check_input_bound(self) -> None:
        """Synchronisation point for CUDA-tensor callers of extract_batch: waits for the queued forwards and raises
        ValueError if any of them violated ``input_bound``."""
        self._bound_check_poll(wait=True)

    def close_session(self):                 # facerec_test.py:124-125
        try:
            if getattr(self.engine, "_h", None):
                self._bound_check_poll(wait=True)
        finally:
            pool = getattr(self, "_decode_pool", None)
            if pool is not None:
                pool.close()
                self._decode_pool = None
            self.engine.close()

    # ---- batched entries (new) ------------------------------------------------------------------
    def extract_batch(self, x):
        """x: float32 [n, h, w, 3] NHWC, preprocessed as preprocess_image leaves it; a CUDA tensor
        (returned: CUDA tensor [n, D], asynchronous) or a NumPy array (returned: NumPy array).  Values must respect
        ``input_bound``: NumPy input is checked here and raises ValueError at once; a CUDA tensor is checked ON THE DEVICE by
        the first kernel, the flag comes back asynchronously and the ValueError is raised by the next call into this object
        after that forward has finished, by ``check_input_bound()`` (waits) or by ``close_session()`` -- read features only
        after one of them when the data's range is not known."""
        torch = _lib.require_gpu()
        self._bound_check_poll()
        if isinstance(x, np.ndarray):
            self._check_bound(x)
            out = []
            for i in range(0, x.shape[0], self.engine.max_batch):
                xd = torch.from_numpy(np.ascontiguousarray(x[i:i + self.engine.max_batch], dtype=np.float32))
                out.append(self.engine.forward(xd.to(self.engine.device), (OUT_FEATURES,))["features"].cpu().numpy())
            return np.concatenate(out) if out else np.zeros((0, self.feature_dim), np.float32)
        out = self.engine.forward(x, (OUT_FEATURES,))["features"]
        self._bound_check_queue()
        return out

    def extract_images(self, imgs_u8, latency: bool = False):
        """Decoded RGB uint8 images [n,H,W,3] (same size; NumPy or CUDA) -> CUDA features [n,D]: the resize +
        BGR + mean of preprocess_image run on the device (bit-exact with the PIL path), then one forward.
        latency: see Engine.forward (extract_features passes it when the extractor was built with latency_plan=True)."""
        from . import preprocess_device
        if self.engine.accepts_u8:      # the resized bytes go straight into the first kernel (no fp32 image in between)
            x8 = preprocess_device.preprocess_pil(imgs_u8, (self.w, self.h), device=self.engine.device, raw_u8=True)
            return self.engine.forward_u8(x8, (OUT_FEATURES,), latency=latency)["features"]
        x = preprocess_device.preprocess_pil(imgs_u8, (self.w, self.h), self.convert2BGR, self.imageNetUtilsMean,
                                             device=self.engine.device)
        return self.engine.forward(x, (OUT_FEATURES,), latency=latency)["features"]

    def extract_files(self, paths: Sequence[str], batch: int = 256, crop_center: bool = False,
                      device_preprocess: bool = True, workers: Optional[int] = None, stats: Optional[dict] = None) -> np.ndarray:
        """The loop of facerec_test.py:394 with the per-image sess.run replaced by batched forwards.

        device_preprocess (default): a PIPELINE -- `workers` decoder PROCESSES (decode_pool.py; default: the cores this
        process may use, at most 32; started once per extractor) write the decoded pixels straight into shared page-locked
        staging slots, two chunks ahead; the slot is copied to the device on a copy stream and resized + mean-subtracted +
        run through the network on the compute stream while the next chunks decode; features come back through pinned
        memory.  The resize is integer arithmetic and a forward does not depend on how images are batched: results are
        bit-identical to the serial per-image path (tests/test_pipeline_gpu.py).
        crop_center or device_preprocess=False: the host does the reference's own preprocessing, image by image.
        stats (optional dict) receives wall seconds of the run and the number of chunks."""
        batch = min(batch, self.engine.max_batch)
        if not (device_preprocess and not crop_center):
            feats: List[np.ndarray] = []
            for i in range(0, len(paths), batch):
                xs = np.stack([self.preprocess_image(p, crop_center) for p in paths[i:i + batch]]).astype(np.float32)
                feats.append(self.extract_batch(xs))
            return np.concatenate(feats) if feats else np.zeros((0, self.feature_dim), np.float32)
        return self._extract_files_pipelined(list(paths), batch, workers, stats)

    def _get_decode_pool(self, workers: Optional[int]):
        from .decode_pool import DecodePool, default_workers
        workers = int(workers or default_workers())
        pool = getattr(self, "_decode_pool", None)
        if pool is not None and pool.workers != workers:
            pool.close()
            pool = None
        if pool is None:
            # one flat byte pool per slot, sized by capacity (256 KiB per image of the largest batch: a 295x295 RGB photo;
            # larger ones spill through the result queue), never one buffer per (H, W)
            pool = self._decode_pool = DecodePool(workers, slot_bytes=max(8 << 20, self.engine.max_batch * (256 << 10)), slots=3)
        return pool

    def _extract_files_pipelined(self, paths: List[str], batch: int, workers: Optional[int], stats: Optional[dict]) -> np.ndarray:
        import time
        from . import preprocess_device
        torch = _lib.require_gpu()
        n = len(paths)
        if n == 0:
            return np.zeros((0, self.feature_dim), np.float32)
        out_host = torch.empty((n, self.feature_dim), dtype=torch.float32).pin_memory()
        dev = self.engine.device
        t0 = time.perf_counter()
        pool = self._get_decode_pool(workers)
        staging = pool.tensor()
        chunks = [(i, min(i + batch, n)) for i in range(0, n, batch)]
        LOOK = pool.slots - 1                                 # chunks decoding ahead of the one being uploaded
        uploaded = [None] * pool.slots                        # event: the copy stream has read staging slot k
        with torch.cuda.device(dev):
            compute = torch.cuda.current_stream(dev)
            copy = torch.cuda.Stream(device=dev)

            def submit(ci):
                if ci < len(chunks):
                    slot = ci % pool.slots
                    if uploaded[slot] is not None:
                        uploaded[slot].synchronize()          # (chunk ci - slots: long done) the slot's bytes are free again
                    lo, hi = chunks[ci]
                    pool.submit(ci, paths[lo:hi], slot)
            try:
                for ci in range(min(LOOK, len(chunks))):
                    submit(ci)
                for ci, (lo, hi) in enumerate(chunks):
                    metas = pool.collect(ci)                  # [(position, (offset, H, W) | None, spilled array | None)]
                    groups = {}                               # (H, W) -> positions, in file order
                    for pos, m, sp in metas:
                        groups.setdefault((m[1], m[2]) if m is not None else sp.shape[:2], []).append(pos)
                    meta_of = {pos: (m, sp) for pos, m, sp in metas}
                    pending = []
                    with torch.cuda.stream(copy):
                        for (H, W), idx in groups.items():
                            nb = H * W * 3
                            d_u8 = torch.empty((len(idx), H, W, 3), dtype=torch.uint8, device=dev)
                            flat = d_u8.view(-1)
                            k = 0
                            while k < len(idx):               # runs of images that are consecutive in staging too -> ONE copy each
                                m, sp = meta_of[idx[k]]
                                if m is None:                 # did not fit its task's region: came back through the queue
                                    flat[k * nb:(k + 1) * nb].copy_(torch.from_numpy(np.ascontiguousarray(sp)).view(-1))
                                    k += 1
                                    continue
                                r = k + 1
                                while r < len(idx) and meta_of[idx[r]][0] is not None and meta_of[idx[r]][0][0] == m[0] + (r - k) * nb:
                                    r += 1
                                flat[k * nb:r * nb].copy_(staging[m[0]:m[0] + (r - k) * nb], non_blocking=True)
                                k = r
                            pending.append((idx, d_u8))
                        up = torch.cuda.Event()
                        up.record(copy)
                    uploaded[ci % pool.slots] = up
                    compute.wait_event(up)
                    for idx, d_u8 in pending:
                        d_u8.record_stream(compute)
                        f = self.extract_images(d_u8)
                        if len(idx) == hi - lo:               # the usual case: one size per chunk -> one contiguous copy back
                            out_host[lo:hi].copy_(f, non_blocking=True)
                        else:
                            out_host[torch.as_tensor(idx, dtype=torch.int64) + lo] = f.cpu()
                    submit(ci + LOOK)
                compute.synchronize()
            except BaseException:
                # leave no half-collected chunk behind: the pool is cheap to restart, stale results are not worth tracking
                pool.close()
                self._decode_pool = None
                raise
        if stats is not None:
            stats["seconds"] = time.perf_counter() - t0
            stats["chunks"] = len(chunks)
            stats["workers"] = pool.workers
            stats["pinned_staging"] = pool.pinned
        return out_host.numpy().copy()


def get_files_of_subjects(db_dir, subjects_file):
    """The 'LFW and YTF concatenation' branch of facerec_test.py:378-380 (README.md:13's LFW-and-YTF row): only the
    sub-directories named in ``subjects_file`` (lfw_ytf_classes.txt: one subject per line), in the file's order.  Inside a
    subject the file names are SORTED, deliberately (as in get_files): the reference takes them in os.walk order, which is the
    directory's on-disk order and differs from machine to machine.  The rows of X / y can therefore be ordered differently from a
    features_file cache the reference wrote on another file system; the labels travel with the rows, so every accuracy is unaffected."""
    with open(subjects_file) as fh:
        subjects = [line.rstrip('\n') for line in fh]
    return [[d, os.path.join(d, f)] for d in subjects if d
            for f in sorted(next(os.walk(os.path.join(db_dir, d)))[2]) if is_image(f)]


def extract_dataset(tfInference, dataset_path: str, features_file: Optional[str] = None, batch: int = 256,
                    crop_center: bool = False, subjects_file: Optional[str] = None):
    """The extract stage of facerec_test.py:377-401: walk ``dataset_path`` (one sub-directory per subject; with
    ``subjects_file`` only the subjects it lists, :378-380), label-encode the directory names, extract every image,
    cache ``np.savez(features_file, x=X, y=y)`` and reuse the cache when the file exists (:308,:399-401).
    Returns (X [N,D] float32, y [N] int)."""
    if features_file is not None and os.path.exists(features_file):
        data = np.load(features_file)
        return data['x'], data['y']
    dirs_and_files = np.array(get_files(dataset_path) if subjects_file is None else get_files_of_subjects(dataset_path, subjects_file))
    dirs = dirs_and_files[:, 0]
    files = dirs_and_files[:, 1]
    classes, y = np.unique(dirs, return_inverse=True)          # == LabelEncoder().fit(dirs).transform(dirs)
    X = tfInference.extract_files([os.path.join(dataset_path, f) for f in files], batch=batch, crop_center=crop_center)
    if features_file is not None:
        np.savez(features_file, x=X, y=y)
    return X, y


def extract_gallery_probe(tfInference, gallery_path: str, probe_path: str, features_file: Optional[str] = None,
                          batch: int = 256, crop_center: bool = False):
    """The extract stage of tf_train_test_recognition (facerec_test.py:220-258): separate Gallery and Probe trees, the label
    encoder FITTED on the gallery's directory names and applied to the probe's (a probe subject the gallery does not have
    raises ValueError, as LabelEncoder.transform does), both trees extracted, cached as
    ``np.savez(features_file, x_train=, y_train=, x_test=, y_test=)`` (:258) and re-used when the file exists (:227).
    Returns (X_train, y_train, X_test, y_test)."""
    if features_file is not None and os.path.exists(features_file):
        data = np.load(features_file)
        return data['x_train'], data['y_train'], data['x_test'], data['y_test']
    train = np.array(get_files(gallery_path))
    classes, y_train = np.unique(train[:, 0], return_inverse=True)       # LabelEncoder().fit(train_dirs) / .transform (:235-237)
    test = np.array(get_files(probe_path))
    pos = np.searchsorted(classes, test[:, 0])
    known = (pos < len(classes)) & (classes[np.minimum(pos, len(classes) - 1)] == test[:, 0])
    if not known.all():
        raise ValueError("y contains previously unseen labels: %r" % sorted(set(test[~known, 0]))[:5])     # :249
    y_test = pos.astype(y_train.dtype)
    X_train = tfInference.extract_files([os.path.join(gallery_path, f) for f in train[:, 1]], batch=batch, crop_center=crop_center)
    X_test = tfInference.extract_files([os.path.join(probe_path, f) for f in test[:, 1]], batch=batch, crop_center=crop_center)
    if features_file is not None:
        np.savez(features_file, x_train=X_train, y_train=y_train, x_test=X_test, y_test=y_test)
    return X_train, y_train, X_test, y_test


_MODELS_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models")
AGE_GENDER_PB = os.path.join(_MODELS_DIR, "age_gender_tf2_new-01-0.14-0.92_quantized.pb")


def get_tf_face_recognizer(model: str = "age_gender", models_dir: Optional[str] = None, **kw) -> TensorFlowInference:
    """The registry of facerec_test.py:209-218 as a function of a name instead of (un)commenting.
    Only 'age_gender' ships with the repository; the VGGFace2 MobileNet/ResNet files are the
    reference's missing blobs (.MISSING_LARGE_BLOBS) and load when a user supplies them."""
    d = models_dir or _MODELS_DIR
    if model == "age_gender":            # facerec_test.py:210
        return TensorFlowInference(os.path.join(d, os.path.basename(AGE_GENDER_PB)), input_tensor='input_1:0',
                                   output_tensor='global_pooling/Mean:0', convert2BGR=True, imageNetUtilsMean=True, **kw)
    if model == "vgg2_mobilenet":        # facerec_test.py:212
        return TensorFlowInference(os.path.join(d, 'vgg2_mobilenet.pb'), input_tensor='input_1:0',
                                   output_tensor='reshape_1/Reshape:0',
                                   learning_phase_tensor='conv1_bn/keras_learning_phase:0', convert2BGR=True,
                                   imageNetUtilsMean=True, **kw)
    if model == "vgg2_mobilenet_h5":     # facerec_test.py:322-334: the same network from its Keras weight file (sz = 192)
        kw.setdefault("input_size", (192, 192))
        return TensorFlowInference(os.path.join(d, 'vgg2_mobilenet.h5'), input_tensor='input_1:0', output_tensor='reshape_1/Reshape:0',
                                   convert2BGR=True, imageNetUtilsMean=True, **kw)
    if model == "vgg2_resnet":           # facerec_test.py:213
        return TensorFlowInference(os.path.join(d, 'vgg2_resnet.pb'), input_tensor='input:0',
                                   output_tensor='pool5_7x7_s1:0', convert2BGR=True, imageNetUtilsMean=False, **kw)
    raise KeyError(model)
