#!/usr/bin/env python3
"""In-network A/B of development-library knobs on one box, one process: the same engine, the knob flipped between timed blocks.
usage: HSEFR_LIB=libhsefr_dev.so [KN_BATCH=n (resnet50)] python tools/knob_net.py <resnet50|agegender|mobilenet192> <knob> <v0,v1,...> [first_op last_op]
prints ms / step per value (three alternating rounds, best of each) and the per-op times of ops first..last."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hse_facerec_tf_amd import _lib, lowering, resnet50
from hse_facerec_tf_amd.engine import Engine
from hse_facerec_tf_amd.graphdef import read_graph
from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB

what, knob = sys.argv[1], sys.argv[2].encode()
vals = [int(v) for v in sys.argv[3].split(",")]
lo, hi = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (0, -1)
rs = np.random.RandomState(123)
if what == "resnet50":
    B, want = int(os.environ.get("KN_BATCH", "128")), (0,)
    plan = resnet50.build_plan(resnet50.synthetic_weights(123), (224, 224), "caffe")
    x = torch.from_numpy(rs.uniform(-128, 128, (B, 224, 224, 3)).astype(np.float32)).cuda()
elif what == "agegender":
    B, want = 512, (0, 1, 2)
    plan = lowering.lower_graph(read_graph(AGE_GENDER_PB), "input_1:0", {0: "global_pooling/Mean:0", 1: "age_pred/Softmax:0", 2: "gender_pred/Sigmoid:0"}, input_bound=256.0)
    x = torch.from_numpy(rs.uniform(-128, 128, (B, 224, 224, 3)).astype(np.float32)).cuda()
else:
    B, want = 256, (0,)
    plan = lowering.lower_graph(read_graph(AGE_GENDER_PB), "input_1:0", {0: "global_pooling/Mean:0"}, (192, 192), input_bound=256.0)
    x = torch.from_numpy(rs.uniform(-128, 128, (B, 192, 192, 3)).astype(np.float32)).cuda()
eng = Engine(plan, max_batch=B)
L = _lib.lib()


def block(steps=30):
    for _ in range(5):
        eng.forward(x, want)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward(x, want)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


best = {v: 1e9 for v in vals}
for rnd in range(3):
    for v in vals:
        _lib.check(L.hsefr_debug_set(knob, v))
        best[v] = min(best[v], block())
for v in vals:
    _lib.check(L.hsefr_debug_set(knob, v))
    line = "%s = %d: %.4f ms/step  %.0f faces/s" % (knob.decode(), v, best[v], B / best[v] * 1e3)
    if hi >= lo:
        eng.set_profiling(20)
        for _ in range(20):
            eng.forward(x, want)
        per = np.mean([eng.op_times_ms(s) for s in range(20)], axis=0)
        eng.set_profiling(0)
        line += "   ops %d..%d us: %s" % (lo, hi, " ".join("%.1f" % (per[i] * 1e3) for i in range(lo, hi + 1)))
    print(line)
