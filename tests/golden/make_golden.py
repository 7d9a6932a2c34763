"""Generates the committed golden vectors under tests/golden/ from the ORACLE (fp64 NumPy graph
interpreter over the reference's own frozen graph) -- run once, here, on CPU:

    python tests/golden/make_golden.py

Inputs (data copied from the reference checkout, Apache-2.0; sha256 recorded in each fixture):
    models/age_gender_tf2_new-01-0.14-0.92_quantized.pb   <- /root/reference/age_gender_identity/
    tests/golden/test_image.jpg                           <- /root/reference/age_gender_identity/

These are RESTATEMENT outputs, not TensorFlow outputs (TensorFlow is not installable here and
the reference holds no golden vectors for this path): parity is "unpinned" -- see oracle/__init__.py.
The 1-NN fixture is different: it is produced by scikit-learn itself (the reference's own library).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import identification as oid          # noqa: E402
from oracle import pipeline as opl                 # noqa: E402
from oracle import tf_graph as tfo                 # noqa: E402

PB = os.path.join(ROOT, "models", "age_gender_tf2_new-01-0.14-0.92_quantized.pb")
IMG = os.path.join(HERE, "test_image.jpg")
FETCH = ["global_pooling/Mean:0", "age_pred/Softmax:0", "gender_pred/Sigmoid:0"]


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def e2e_image():
    g = tfo.GraphOracle(PB, np.float64)
    img = opl.imread_rgb(IMG)
    out = {"pb_sha256": sha(PB), "img_sha256": sha(IMG)}
    for s in (224, 192):
        # facerec_test.py path: PIL-BILINEAR whole image -> s x s, BGR - ImageNet mean (float64)
        x = opl.preprocess_image(img, s, s, True, True)[None]
        f, a, ge = g.run(FETCH, {"input_1:0": x})
        out["feat_%d" % s], out["age_%d" % s], out["gender_%d" % s] = f[0].astype(np.float32), a[0].astype(np.float32), ge[0].astype(np.float32)
    # facial_analysis.py path (cv2-style resize, float32 preprocessing), whole frame as the "face"
    x = opl.age_gender_preprocess(img, 224, 224)
    f, a, ge = g.run(FETCH, {"input_1:0": x})
    out["ag_feat"], out["ag_age"], out["ag_gender"] = f[0].astype(np.float32), a[0].astype(np.float32), ge[0].astype(np.float32)
    out["ag_res_age"] = np.float64(opl.decode_age(a[0])[0])
    # four fixed crops of the frame (stand-ins for detections), through process_image's box geometry
    boxes = np.array([[100, 60, 260, 260], [330, 120, 470, 300], [520, 40, 700, 280], [-5, 400, 150, 600]])
    out["boxes"] = boxes
    feats, ages, genders = [], [], []
    for (x1, y1, x2, y2) in boxes:
        x1, x2, y1, y2 = max(x1 - 10, 0), min(x2 + 10, img.shape[1]), max(y1 - 10, 0), min(y2 + 10, img.shape[0])
        xx = opl.age_gender_preprocess(img[y1:y2, x1:x2], 224, 224)
        f, a, ge = g.run(FETCH, {"input_1:0": xx})
        feats.append(f[0]); ages.append(opl.decode_age(a[0])[0]); genders.append(ge[0])
    out["crop_feats"] = np.asarray(feats, np.float32)
    out["crop_ages"] = np.asarray(ages, np.float64)
    out["crop_genders"] = np.asarray(genders, np.float32)
    np.savez_compressed(os.path.join(HERE, "e2e_test_image.npz"), **out)
    print("e2e_test_image: argsort(age)[-2:] =", np.argsort(out["age_224"])[-2:], "gender", out["gender_224"],
          "|f| =", np.linalg.norm(out["feat_224"]))


def e2e_synthetic():
    g = tfo.GraphOracle(PB, np.float64)
    out = {"pb_sha256": sha(PB), "seed": 123}
    for s, n in ((192, 3), (224, 2), (96, 3), (100, 2)):
        # bench.py's input distribution: RandomState(123), U(-128,128) fp32 NHWC
        x = np.random.RandomState(123).uniform(-128, 128, (n, s, s, 3)).astype(np.float32)
        f, a, ge = g.run(FETCH, {"input_1:0": x})
        out["feat_%d" % s], out["age_%d" % s], out["gender_%d" % s] = f.astype(np.float32), a.astype(np.float32), ge.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "e2e_synthetic.npz"), **out)
    print("e2e_synthetic done")


def kernels():
    """Per-kernel-class I/O pairs on seeded tensors (odd sizes, both strides, ragged M tails)."""
    rs = np.random.RandomState(7)
    out = {}
    f64 = np.float64

    def act6(v):
        return np.minimum(np.maximum(v, 0), 6)

    # depthwise: (n,h,w,c,stride)
    for i, (n, h, w, c, s) in enumerate([(2, 9, 7, 8, 1), (1, 10, 10, 32, 2), (2, 7, 9, 16, 2), (1, 6, 6, 64, 1),
                                          (1, 1, 1, 4, 1), (1, 2, 3, 4, 2)]):
        x = rs.uniform(-3, 6, (n, h, w, c)).astype(np.float32)
        k = rs.randn(3, 3, c, 1).astype(np.float32)
        sc = rs.uniform(0.5, 2, c).astype(np.float32)
        sh = rs.randn(c).astype(np.float32)
        y = act6(tfo.depthwise_conv2d(x.astype(f64), k, (s, s), "SAME") * sc + sh)
        out.update({"dw%d_x" % i: x, "dw%d_k" % i: k, "dw%d_sc" % i: sc, "dw%d_sh" % i: sh, "dw%d_s" % i: s,
                    "dw%d_y" % i: y.astype(np.float32)})
    # first conv
    for i, (n, h, w, cout, s) in enumerate([(2, 12, 12, 32, 2), (1, 9, 11, 32, 2), (1, 8, 6, 8, 1), (1, 1, 1, 4, 2)]):
        x = rs.uniform(-128, 128, (n, h, w, 3)).astype(np.float32)
        k = (rs.randn(3, 3, 3, cout) * 0.05).astype(np.float32)
        sh = rs.randn(cout).astype(np.float32)
        y = act6(tfo.conv2d(x.astype(f64), k, (s, s), "SAME") + sh)
        out.update({"c3%d_x" % i: x, "c3%d_k" % i: k, "c3%d_sh" % i: sh, "c3%d_s" % i: s, "c3%d_y" % i: y.astype(np.float32)})
    # pointwise: (m, k, cout)
    for i, (m, kk, cout) in enumerate([(128, 32, 64), (200, 64, 128), (37, 128, 256), (1, 32, 64), (300, 96, 192)]):
        x = rs.uniform(0, 6, (m, kk)).astype(np.float32)
        k = (rs.randn(kk, cout) / np.sqrt(kk)).astype(np.float32)
        sh = rs.randn(cout).astype(np.float32)
        y = act6(x.astype(f64).dot(k.astype(f64)) + sh)
        out.update({"pw%d_x" % i: x, "pw%d_k" % i: k, "pw%d_sh" % i: sh, "pw%d_y" % i: y.astype(np.float32)})
    # gap / dense / softmax
    x = rs.uniform(0, 6, (3, 6, 6, 64)).astype(np.float32)
    out.update({"gap_x": x, "gap_y": x.astype(f64).mean(axis=(1, 2)).astype(np.float32)})
    x = rs.randn(11, 96).astype(np.float32)
    k = rs.randn(96, 100).astype(np.float32) * 0.2
    b = rs.randn(100).astype(np.float32)
    z = x.astype(f64).dot(k) + b
    out.update({"dn_x": x, "dn_k": k, "dn_b": b, "dn_y_none": z.astype(np.float32),
                "dn_y_relu": np.maximum(z, 0).astype(np.float32), "dn_y_sigmoid": tfo.sigmoid(z).astype(np.float32),
                "sm_y": tfo.softmax(z).astype(np.float32)})
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **out)
    print("kernels done")


def nn1():
    X, y = oid.synthetic_gallery(n_classes=150, dim=256, seed=123, noise=1.6)
    Xn, y2, kept = oid.filter_and_encode(X, y)
    acc, train, test, y_pred, nn_idx, nn_dist = oid.one_nn(Xn, y2)
    np.savez_compressed(os.path.join(HERE, "nn1.npz"), n_classes=150, dim=256, seed=123, noise=1.6, kept=kept, y=y2,
                        train=train, test=test, y_pred=y_pred, nn_index=nn_idx, nn_dist=nn_dist.astype(np.float32),
                        accuracy=acc, x_norm_sample=Xn[:8].astype(np.float32))
    print("nn1: N=%d kept=%d classes=%d acc=%.4f" % (len(y), len(y2), y2.max() + 1, acc))


def nn1_lfw():
    """BASELINE configs[4] at full size: 9164 embeddings of 1680 classes (the product's LFW-shaped label vector) ->
    scikit-learn's own normalize / StratifiedShuffleSplit / KNeighborsClassifier(1) -> 4582 x 4582 x 1024."""
    from hse_facerec_tf_amd import gallery
    y = gallery.lfw_like_labels(9164, 1680)
    noise = 2.0
    X = oid.embeddings_for_labels(y, 1024, 123, noise)
    Xn, y2, kept = oid.filter_and_encode(X, y)
    acc, train, test, y_pred, nn_idx, nn_dist = oid.one_nn(Xn, y2)
    # margin between the nearest and the second-nearest gallery row of every probe (fp64): the fixture must not hinge on ties
    d2 = 2.0 - 2.0 * (Xn[test].astype(np.float64) @ Xn[train].astype(np.float64).T)
    part = np.partition(d2, 1, axis=1)
    margin = float((part[:, 1] - part[:, 0]).min())
    np.savez_compressed(os.path.join(HERE, "nn1_lfw.npz"), n=9164, n_classes=1680, dim=1024, seed=123, noise=noise,
                        train=train.astype(np.int32), test=test.astype(np.int32), nn_index=nn_idx.astype(np.int32),
                        y_pred=y_pred.astype(np.int32), accuracy=acc, min_margin=margin, kept_all=bool(len(kept) == 9164))
    print("nn1_lfw: kept=%d acc=%.6f min d2 margin=%.3e" % (len(kept), acc, margin))


def protocols():
    """The two other identification protocols of facerec_test.py, frozen from NumPy / scikit-learn themselves:
    get_single_image_per_class_cv (:177-197) -> cross_validate(KNeighborsClassifier(1)) (:199-207), and the gallery / probe
    split of tf_train_test_recognition (:220-288)."""
    X, y = oid.synthetic_gallery(n_classes=120, dim=256, seed=321, noise=1.5)
    Xn, y2, kept = oid.filter_and_encode(X, y)
    cv = oid.single_image_per_class_cv(y2, n_splits=10, random_state=0)
    accs = oid.cross_validate_1nn(Xn, y2, cv)
    # gallery / probe: even samples of every class are the gallery tree, odd ones the probe tree; features NOT normalised
    Xk = X[kept]
    order = np.argsort(y2, kind="stable")
    pos_in_class = np.zeros(len(y2), dtype=np.int64)
    for c in np.unique(y2):
        m = order[y2[order] == c]
        pos_in_class[m] = np.arange(len(m))
    g, p = np.nonzero(pos_in_class % 2 == 0)[0], np.nonzero(pos_in_class % 2 == 1)[0]
    acc_gp, pred_gp, idx_gp, dist_gp = oid.gallery_probe_1nn(Xk[g], y2[g], Xk[p], y2[p])
    acc_gpn, pred_gpn, idx_gpn, _ = oid.gallery_probe_1nn(Xn[g], y2[g], Xn[p], y2[p])
    out = {"n_classes": 120, "dim": 256, "seed": 321, "noise": 1.5, "y": y2, "accuracies": accs,
           "gallery": g, "probe": p, "gp_accuracy_percent": acc_gp, "gp_pred": pred_gp, "gp_nn_index": idx_gp,
           "gp_nn_dist": dist_gp.astype(np.float32), "gpn_accuracy_percent": acc_gpn, "gpn_pred": pred_gpn, "gpn_nn_index": idx_gpn}
    for i, (tr, te) in enumerate(cv):
        out["train_%d" % i], out["test_%d" % i] = tr, te
    np.savez_compressed(os.path.join(HERE, "protocols.npz"), **out)
    print("protocols: single-image accs mean %.4f std %.4f; gallery/probe %.2f %% (raw) %.2f %% (normalised)"
          % (accs.mean(), accs.std(), acc_gp, acc_gpn))


def mtcnn():
    """Oracle MTCNN cascade (fp32 graph interpreter + restated INTER_AREA) on the reference's demo image, then the
    oracle age/gender model on the detected faces exactly as process_image crops them (facial_analysis.py:233-271)."""
    from oracle.mtcnn import OracleMTCNN
    MT = os.path.join(ROOT, "models", "mtcnn.pb")
    img = opl.imread_rgb(IMG)
    det = OracleMTCNN(MT, minsize=32, compute_dtype=np.float32)
    boxes, points = det.detect(img)
    ag = opl.OracleAgeGender(PB, np.float64)
    ages, genders, feats = [], [], []
    for b in boxes:
        x1, y1, x2, y2 = [int(v) for v in b[:4]]
        x1, x2, y1, y2 = max(x1 - 10, 0), min(x2 + 10, img.shape[1]), max(y1 - 10, 0), min(y2 + 10, img.shape[0])
        a, g_, f = ag.age_gender_fun(img[y1:y2, x1:x2])
        ages.append(a); genders.append(g_); feats.append(f)
    rs = np.random.RandomState(5)
    nets = {}
    for name, shape, outs in (("pnet", (1, 37, 53, 3), ['pnet/conv4-2/BiasAdd:0', 'pnet/prob1:0']),
                              ("rnet", (5, 24, 24, 3), ['rnet/conv5-2/conv5-2:0', 'rnet/prob1:0']),
                              ("onet", (4, 48, 48, 3), ['onet/conv6-2/conv6-2:0', 'onet/conv6-3/conv6-3:0', 'onet/prob1:0'])):
        x = rs.uniform(-1, 1, shape).astype(np.float32)
        r = tfo.GraphOracle(MT, np.float64).run(outs, {name + "/input:0": x})
        nets[name + "_x"] = x
        for i, a in enumerate(r):
            nets["%s_out%d" % (name, i)] = np.asarray(a, np.float32)
    np.savez_compressed(os.path.join(HERE, "mtcnn_test_image.npz"), mtcnn_sha256=sha(MT), img_sha256=sha(IMG), boxes=boxes,
                        points=points, ages=np.asarray(ages, np.float64), genders=np.asarray(genders, np.float32),
                        feats=np.asarray(feats, np.float32), **nets)
    print("mtcnn: %d faces" % boxes.shape[0], np.round(boxes, 2).tolist(), "ages", np.round(ages, 1))


if __name__ == "__main__":
    which = sys.argv[1:] or ["e2e_image", "e2e_synthetic", "kernels", "nn1", "nn1_lfw", "protocols", "mtcnn"]
    for w in which:
        globals()[w]()
