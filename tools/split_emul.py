"""Accuracy study (CPU emulation): pointwise layers computed as a bf16 split product on top of an fp32 accumulate."""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import plan_ref
from hse_facerec_tf_amd import graphdef, lowering

def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)

def split(x, parts):
    out = []; rem = x.astype(np.float32)
    for _ in range(parts):
        h = bf16(rem); out.append(h); rem = (rem - h).astype(np.float32)
    return out

def make_pw(MODE):
    def pw(a, wt):
        a = a.astype(np.float32); wt = wt.astype(np.float32)
        if MODE == "f32": return a.dot(wt.T)
        if MODE == "bf16": return bf16(a).dot(bf16(wt).T)
        np_ = 2 if MODE == "x3" else 3
        A = split(a, np_); W = split(wt, np_)
        acc = np.zeros((a.shape[0], wt.shape[0]), np.float32)
        pairs = [(0,0),(0,1),(1,0)] if MODE == "x3" else [(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)]
        for i, j in reversed(pairs): acc += A[i].dot(W[j].T)
        return acc
    return pw

MODEL_PB = "/root/repo/models/age_gender_tf2_new-01-0.14-0.92_quantized.pb"
GOLDEN = "/root/repo/tests/golden"
g = graphdef.read_graph(MODEL_PB)
rel = lambda a, b: float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / np.abs(b).max())
z = np.load(os.path.join(GOLDEN, "e2e_synthetic.npz"))
for size in (96, 192):
    plan = lowering.lower_graph(g, "input_1:0", {0: "global_pooling/Mean:0", 1: "age_pred/Softmax:0", 2: "gender_pred/Sigmoid:0"}, (size, size))
    n = z["feat_%d" % size].shape[0]
    x = np.random.RandomState(123).uniform(-128, 128, (n, size, size, 3)).astype(np.float32)
    for MODE in ("f32", "x6", "x3", "bf16"):
        plan_ref.PW = make_pw(MODE)
        out = plan_ref.run(plan.serialize(), x, dtype=np.float32)
        f, gg = out["features"].astype(np.float64), z["feat_%d" % size].astype(np.float64)
        big = np.abs(gg) > 1e-3 * np.abs(gg).max()
        print(MODE, size, "feat %.2e age %.2e gender %.2e  elemwise %.2e" % (rel(out["features"], z["feat_%d" % size]), rel(out["age_probs"], z["age_%d" % size]), rel(out["gender"], z["gender_%d" % size]), (np.abs(f - gg)[big] / np.abs(gg)[big]).max()))
