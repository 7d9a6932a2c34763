"""Oracle part 4: the post-extract identification protocol, run on scikit-learn itself.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows facerec_test.py:401-414 (L2-normalise, keep classes with more than one image,
re-encode labels) and facerec_test.py:200-207 / :430 (``StratifiedShuffleSplit(n_splits=1,
test_size=0.5, random_state=0)`` + ``KNeighborsClassifier(n_neighbors=1, p=2)`` accuracy).
scikit-learn IS importable on this image, so this leg is the reference's own library
calls, not a restatement; ``tests/golden/make_golden.py`` freezes its split indices and
predictions into fixtures.
"""
from __future__ import annotations

import numpy as np


def filter_and_encode(X: np.ndarray, y: np.ndarray):
    """facerec_test.py:403-414."""
    from sklearn import preprocessing
    X_norm = preprocessing.normalize(X, norm="l2")
    y_l = list(y)
    counts = {}
    for el in y_l:
        counts[el] = counts.get(el, 0) + 1
    indices = [i for i, el in enumerate(y_l) if counts[el] > 1]   # == y_l.count(el) > 1, without the O(N^2)
    y2 = np.asarray(y)[indices]
    enc = preprocessing.LabelEncoder()
    enc.fit(y2)
    y2 = enc.transform(y2)
    return X_norm[indices, :], y2, np.asarray(indices)


def split_indices(X_norm: np.ndarray, y: np.ndarray):
    """facerec_test.py:202: the one stratified 50/50 split, random_state=0."""
    from sklearn import model_selection
    sss = model_selection.StratifiedShuffleSplit(n_splits=1, test_size=0.5, random_state=0)
    (train, test), = sss.split(X_norm, y)
    return train, test


def one_nn(X_norm: np.ndarray, y: np.ndarray):
    """facerec_test.py:200-207 with the classifier of :422 -> (accuracy, train, test, y_pred, nn_index)."""
    from sklearn.neighbors import KNeighborsClassifier
    train, test = split_indices(X_norm, y)
    clf = KNeighborsClassifier(n_neighbors=1, p=2)
    clf.fit(X_norm[train], y[train])
    dist, idx = clf.kneighbors(X_norm[test], n_neighbors=1)
    y_pred = y[train][idx[:, 0]]
    acc = float((y_pred == y[test]).mean())
    return acc, train, test, y_pred, idx[:, 0], dist[:, 0]


def synthetic_gallery(n_classes: int = 200, dim: int = 1024, seed: int = 123, noise: float = 0.35):
    """LFW-shaped synthetic embeddings: Gaussian class centroids, a
    long-tailed class-size histogram with singletons (which the protocol must drop),
    non-negative like post-ReLU6 GAP features."""
    rs = np.random.RandomState(seed)
    sizes = np.maximum(1, np.round(rs.pareto(1.2, n_classes) + 1).astype(int))
    sizes = np.minimum(sizes, 40)
    cent = rs.randn(n_classes, dim).astype(np.float32)
    X, y = [], []
    for c, s in enumerate(sizes):
        X.append(np.maximum(cent[c] + noise * rs.randn(s, dim).astype(np.float32), 0))
        y += [c] * s
    X = np.concatenate(X).astype(np.float32)
    y = np.asarray(y)
    perm = rs.permutation(len(y))
    return X[perm], y[perm]


def embeddings_for_labels(y: np.ndarray, dim: int = 1024, seed: int = 123, noise: float = 1.0) -> np.ndarray:
    """Config-5-sized synthetic embeddings for a GIVEN label vector (directory-walk order): Gaussian class
    centroids + isotropic noise, clipped at zero like post-ReLU6 GAP features.  Deterministic in (y, dim, seed,
    noise); drawn class by class so that it does not depend on NumPy's block sizes."""
    rs = np.random.RandomState(seed)
    y = np.asarray(y)
    n_classes = int(y.max()) + 1
    cent = rs.randn(n_classes, dim).astype(np.float32)
    X = np.empty((len(y), dim), np.float32)
    for c in range(n_classes):
        rows = np.nonzero(y == c)[0]
        X[rows] = np.maximum(cent[c] + noise * rs.randn(len(rows), dim).astype(np.float32), 0)
    return X


def single_image_per_class_cv(y: np.ndarray, n_splits: int = 10, random_state: int = 0):
    """facerec_test.py:177-197 as written there: NumPy's GLOBAL generator seeded once, then -- split after split, class
    after class in np.unique order -- the class's indices shuffled in place, the first kept for training, the rest for
    testing.  The caller's global generator state is put back afterwards (the only departure: an oracle must not disturb
    the tests around it)."""
    saved = np.random.get_state()
    try:
        res_cv = []
        inds = np.arange(len(y))
        np.random.seed(random_state)
        for _ in range(n_splits):
            inds_train, inds_test = [], []
            for lbl in np.unique(y):
                tmp_inds = inds[y == lbl]
                np.random.shuffle(tmp_inds)
                last_ind = 1
                inds_train.extend(tmp_inds[:last_ind])
                inds_test.extend(tmp_inds[last_ind:])
            res_cv.append((np.array(inds_train), np.array(inds_test)))
        return res_cv
    finally:
        np.random.set_state(saved)


def cross_validate_1nn(X_norm: np.ndarray, y: np.ndarray, cv):
    """classifier_tester (facerec_test.py:199-207) with the commented-in ``sss=get_single_image_per_class_cv(y)``:
    scikit-learn's own cross_validate over explicit splits.  Returns the test accuracies."""
    from sklearn import model_selection
    from sklearn.neighbors import KNeighborsClassifier
    scores = model_selection.cross_validate(KNeighborsClassifier(n_neighbors=1, p=2), X_norm, y, scoring="accuracy", cv=cv)
    return scores["test_score"]


def gallery_probe_1nn(X_train: np.ndarray, y_train: np.ndarray, X_test: np.ndarray, y_test: np.ndarray):
    """facerec_test.py:271,282-288: KNeighborsClassifier(n_neighbors=1, p=2).fit(X_train, y_train).predict(X_test) --
    on the features AS LOADED (the reference normalises into X_train_norm / X_test_norm at :262,265 and then does not use
    them) -- accuracy in percent as printed there, plus the nearest gallery indices."""
    from sklearn.neighbors import KNeighborsClassifier
    clf = KNeighborsClassifier(n_neighbors=1, p=2)
    clf.fit(X_train, y_train)
    y_test_pred = clf.predict(X_test)
    acc = 100.0 * (y_test == y_test_pred).sum() / len(y_test)
    dist, idx = clf.kneighbors(X_test, n_neighbors=1)
    return acc, y_test_pred, idx[:, 0], dist[:, 0]
