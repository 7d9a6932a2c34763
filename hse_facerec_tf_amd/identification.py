"""Post-extract identification (facerec_test.py:401-432, 200-207) with the distance work on the GPU.

Host steps stay what the reference does on the host -- including its own scikit-learn calls
for the label encoding and the stratified split (scikit-learn is a dependency of the
reference's callers, not of the engine); the O(N_test x N_train x D) nearest-neighbour search
(KNeighborsClassifier(1).fit/predict, :422,:203) and the L2 normalisation (:401) run through
libhsefr (ops.l2_normalize / ops.nn1).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np


def filter_classes(y: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """facerec_test.py:407-412: keep samples whose label occurs more than once, re-encode the
    labels to 0..C-1 in sorted order (LabelEncoder).  Returns (indices, y_encoded)."""
    y = np.asarray(y)
    classes, inverse, counts = np.unique(y, return_inverse=True, return_counts=True)
    indices = np.nonzero(counts[inverse] > 1)[0]
    _, y_enc = np.unique(y[indices], return_inverse=True)
    return indices, y_enc


def stratified_half_split(y: np.ndarray, random_state: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """StratifiedShuffleSplit(n_splits=1, test_size=0.5, random_state=0) (facerec_test.py:202)."""
    from sklearn import model_selection
    sss = model_selection.StratifiedShuffleSplit(n_splits=1, test_size=0.5, random_state=random_state)
    (train, test), = sss.split(np.zeros((len(y), 1)), y)
    return train, test


class SplitJob:
    """The host half of the protocol -- filter_classes + stratified_half_split, 6-9 ms of scikit-learn for LFW's 9164 labels --
    running in a thread of its own.  It depends on the labels only, which the dataset walk yields before the first image is
    decoded (facerec_test.py:377-392), so a caller can start it while the device is still extracting (gallery.extract_sharded's
    ``on_issued`` hook) and hand it to one_nn_identification(split=job): same indices, same split, off the critical path."""

    def __init__(self, y: np.ndarray, random_state: int = 0):
        import threading
        self._out = None
        self._err = None

        def work():
            try:
                indices, y_enc = filter_classes(y)
                train, test = stratified_half_split(y_enc, random_state)
                self._out = (indices, y_enc, train, test)
            except BaseException as e:          # re-raised in the caller's thread
                self._err = e
        self._thread = threading.Thread(target=work, name="hsefr-split", daemon=True)
        self._thread.start()

    def result(self):
        self._thread.join()
        if self._err is not None:
            raise self._err
        return self._out


def start_split(y: np.ndarray, random_state: int = 0) -> SplitJob:
    return SplitJob(np.asarray(y), random_state)


def one_nn_identification(X, y: np.ndarray, split=None,
                          pca_components: Optional[int] = None, timings: Optional[dict] = None, device=None) -> Dict:
    """The protocol of facerec_test.py:401-432: 'k-NN' (pca_components=None) or 'k-NN+PCA'
    (pca_components=128, the Pipeline of :421 -- PCA is fitted on the gallery half by scikit-learn on
    the host, exactly as the reference does, and the projected vectors go back to the device for the search).

    X: [N, D] float32 embeddings, CUDA tensor or NumPy array (uploaded to ``device``, default: the current one); y: [N] labels.
    split: None (compute it here), a (train, test) pair over the FILTERED samples, or a SplitJob started on the same labels.
    Returns accuracy, the split, predictions and nearest-gallery indices.  ``timings`` (optional dict) receives the
    device-synchronised wall seconds of each phase: normalize_s, host_split_s, select_s, nn1_s, readback_s (indices and
    distances back to the host + the label comparison)."""
    import time
    from . import _lib, ops
    torch = _lib.require_gpu()
    if isinstance(X, np.ndarray):
        X = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(_lib.cuda_device(device))

    def lap(key, t_prev):
        if timings is None:
            return 0.0
        torch.cuda.synchronize(X.device)
        t = time.perf_counter()
        timings[key] = t - t_prev
        return t

    t = lap("_start", 0.0)
    Xn = ops.l2_normalize(X.contiguous())                       # :401
    t = lap("normalize_s", t)
    if isinstance(split, SplitJob):                             # started earlier on the same labels: wait for it
        indices, y_enc, train, test = split.result()
    else:
        indices, y_enc = filter_classes(y)                      # :407-412
        train, test = split if split is not None else stratified_half_split(y_enc)
    t = lap("host_split_s", t)
    Xn = Xn[torch.from_numpy(indices).to(Xn.device)].contiguous()   # :413
    gal = Xn[torch.from_numpy(train).to(Xn.device)].contiguous()
    qry = Xn[torch.from_numpy(test).to(Xn.device)].contiguous()
    t = lap("select_s", t)
    if pca_components:
        from sklearn.decomposition import PCA
        pca = PCA(n_components=pca_components).fit(gal.cpu().numpy())
        pad = (-pca_components) % 8                      # hsefr_nn1 wants d % 8 == 0: zero columns change no distance

        def proj(t):
            z = pca.transform(t.cpu().numpy()).astype(np.float32)
            return torch.from_numpy(np.pad(z, ((0, 0), (0, pad)))).to(Xn.device).contiguous()
        gal, qry = proj(gal), proj(qry)
    nn_idx, nn_d2 = ops.nn1(qry, gal)
    t = lap("nn1_s", t)
    nn_idx_h = nn_idx.cpu().numpy()
    nn_dist_h = np.sqrt(nn_d2.cpu().numpy())
    y_pred = y_enc[train][nn_idx_h]
    acc = float((y_pred == y_enc[test]).mean()) if len(test) else float("nan")
    t = lap("readback_s", t)
    if timings is not None:
        timings.pop("_start", None)
        timings["nn1_shape"] = (int(qry.shape[0]), int(gal.shape[0]), int(qry.shape[1]))
    return {"accuracy": acc, "indices": indices, "y": y_enc, "train": train, "test": test, "y_pred": y_pred,
            "nn_index": nn_idx_h, "nn_dist": nn_dist_h, "num_classes": int(y_enc.max() + 1) if len(y_enc) else 0}


def single_image_per_class_splits(y: np.ndarray, n_splits: int = 10, random_state: int = 0):
    """get_single_image_per_class_cv (facerec_test.py:177-197), the protocol behind README.md:13's one-training-image rows:
    ``n_splits`` splits; in each, every class's sample indices are shuffled and the FIRST one is the gallery image, the rest
    are probes.  The reference seeds NumPy's global generator once (``np.random.seed(random_state)``) and shuffles class by
    class in ``np.unique`` order, split after split; a private ``RandomState(random_state)`` draws the same stream without
    touching the caller's global state -- the splits are bit-equal (tests/golden/protocols.npz, written by
    tests/golden/make_golden.py from the reference's literal globally seeded loop)."""
    y = np.asarray(y)
    inds = np.arange(len(y))
    rs = np.random.RandomState(random_state)
    classes = np.unique(y)
    members = [inds[y == lbl] for lbl in classes]
    res_cv = []
    for _ in range(n_splits):
        inds_train, inds_test = [], []
        for m in members:
            tmp = m.copy()
            rs.shuffle(tmp)
            inds_train.extend(tmp[:1])
            inds_test.extend(tmp[1:])
        res_cv.append((np.array(inds_train), np.array(inds_test)))
    return res_cv


def _nn1_predict(torch, ops, Xd, train, test, y):
    gal = Xd[torch.from_numpy(np.asarray(train, dtype=np.int64)).to(Xd.device)].contiguous()
    qry = Xd[torch.from_numpy(np.asarray(test, dtype=np.int64)).to(Xd.device)].contiguous()
    nn_idx, nn_d2 = ops.nn1(qry, gal)
    nn_idx_h = nn_idx.cpu().numpy()
    return y[np.asarray(train)][nn_idx_h], nn_idx_h, np.sqrt(nn_d2.cpu().numpy())


def cross_validated_1nn(X, y: np.ndarray, cv, normalize: bool = True, device=None) -> Dict:
    """classifier_tester (facerec_test.py:199-207) for KNeighborsClassifier(n_neighbors=1, p=2) over an explicit list of
    (train, test) index pairs -- e.g. single_image_per_class_splits(y) in place of the stratified half split (:200-201) --
    with every search on the GPU.  Returns the per-split accuracies and their mean / std as the reference prints them."""
    from . import _lib, ops
    torch = _lib.require_gpu()
    if isinstance(X, np.ndarray):
        X = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(_lib.cuda_device(device))
    Xd = ops.l2_normalize(X.contiguous()) if normalize else X.contiguous()
    y = np.asarray(y)
    accs, preds = [], []
    for train, test in cv:
        y_pred, _, _ = _nn1_predict(torch, ops, Xd, train, test, y)
        preds.append(y_pred)
        accs.append(float((y_pred == y[np.asarray(test)]).mean()) if len(test) else float("nan"))
    accs = np.asarray(accs)
    return {"accuracies": accs, "mean": float(accs.mean()) if len(accs) else float("nan"),
            "std": float(accs.std()) if len(accs) else float("nan"), "y_pred": preds}


def gallery_probe_identification(X_train, y_train: np.ndarray, X_test, y_test: np.ndarray, normalize: bool = False,
                                 pca_components: Optional[int] = None, device=None) -> Dict:
    """The gallery / probe protocol of tf_train_test_recognition (facerec_test.py:260-288): the '1-NN' classifier (and
    '1-NN+PCA' with ``pca_components``, 16 at :269) FITTED on the gallery features, every probe labelled by its nearest
    gallery row; accuracy = share of probes whose label is right (:287).  NB the reference computes L2-normalised copies
    (:262,265) and then fits / predicts on the UN-normalised ``X_train`` / ``X_test`` (:284-285): ``normalize=False`` is what
    it runs, ``normalize=True`` what the copies suggest it meant.  The search runs on the GPU (hsefr_nn1: ties -> the lowest
    gallery index, scikit-learn's own choice)."""
    from . import _lib, ops
    torch = _lib.require_gpu()
    dev = _lib.cuda_device(device)

    def up(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev) if isinstance(a, np.ndarray) else a.float()
        return ops.l2_normalize(t.contiguous()) if normalize else t.contiguous()
    gal, qry = up(X_train), up(X_test)
    y_train, y_test = np.asarray(y_train), np.asarray(y_test)
    if gal.shape[0] != len(y_train) or qry.shape[0] != len(y_test):
        raise ValueError("features and labels differ in length: %d/%d gallery, %d/%d probe"
                         % (gal.shape[0], len(y_train), qry.shape[0], len(y_test)))
    if pca_components:
        from sklearn.decomposition import PCA
        pca = PCA(n_components=pca_components).fit(gal.cpu().numpy())
        pad = (-pca_components) % 8

        def proj(t):
            z = pca.transform(t.cpu().numpy()).astype(np.float32)
            return torch.from_numpy(np.pad(z, ((0, 0), (0, pad)))).to(dev).contiguous()
        gal, qry = proj(gal), proj(qry)
    nn_idx, nn_d2 = ops.nn1(qry, gal)
    nn_idx_h = nn_idx.cpu().numpy()
    y_pred = y_train[nn_idx_h]
    acc = float((y_pred == y_test).mean()) if len(y_test) else float("nan")
    return {"accuracy": acc, "y_pred": y_pred, "nn_index": nn_idx_h, "nn_dist": np.sqrt(nn_d2.cpu().numpy())}


def feature_distance_matrix(features, born_years=None, photo_years=None, device=None) -> np.ndarray:
    """The dist_matrix of process_photos.perform_clustering (process_photos.py:45-60): Euclidean distance
    between facial features (on the GPU) plus 0.1 x the age term (cur_age_i - cur_age_j)^2 / (cur_age_i +
    cur_age_j), cur_age = max(year_i, year_j) - born_year, clipped at 0.  Returns a host float64 matrix as the
    clustering code (facial_clustering.get_facial_clusters) expects."""
    from . import _lib, ops
    torch = _lib.require_gpu()
    f = torch.from_numpy(np.ascontiguousarray(features, dtype=np.float32)).to(_lib.cuda_device(device)) if isinstance(features, np.ndarray) \
        else features.float().contiguous()
    dist = ops.pairwise_distances(f).cpu().numpy().astype(np.float64)
    if born_years is not None:
        by = np.asarray(born_years, dtype=np.float64)
        yr = np.asarray(photo_years, dtype=np.float64)
        max_year = np.maximum(yr[:, None], yr[None, :])
        ai, aj = max_year - by[:, None], max_year - by[None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            dist = dist + 0.1 * (ai - aj) ** 2 / (ai + aj)
    return np.clip(dist, a_min=0, a_max=None)
