#!/usr/bin/env python3
"""MobileNet-192 batch 256 (or KN_WHAT=agegender: 224 x 224 x 512) with block_fusion = auto | all: ms / step and the per-op table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hse_facerec_tf_amd import lowering
from hse_facerec_tf_amd.engine import Engine
from hse_facerec_tf_amd.graphdef import read_graph
from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB
rs = np.random.RandomState(1)
ag = os.environ.get("KN_WHAT") == "agegender"
B, hw = (512, 224) if ag else (256, 192)
x = torch.from_numpy(rs.uniform(-128, 128, (B, hw, hw, 3)).astype(np.float32)).cuda()
outs = {}
for mode in sys.argv[1:] or ["auto", "all"]:
    plan = lowering.lower_graph(read_graph(AGE_GENDER_PB), "input_1:0", {0: "global_pooling/Mean:0"}, (hw, hw), input_bound=256.0, block_fusion=mode)
    eng = Engine(plan, max_batch=B)
    for _ in range(5): y = eng.forward(x, (0,))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): eng.forward(x, (0,))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    outs[mode] = list(y.values())[0].float().cpu().numpy()
    eng.set_profiling(20)
    for _ in range(20): eng.forward(x, (0,))
    per = np.mean([eng.op_times_ms(s) for s in range(20)], axis=0)
    print("%s: %.4f ms/step %.0f faces/s kinds %s\n   us: %s" % (mode, dt * 1e3, B / dt, [L.kind for L in plan.layers], " ".join("%.1f" % (t * 1e3) for t in per)))
    eng.close()
ks = list(outs)
if len(ks) > 1: print("max |diff| %s vs %s: %.3g" % (ks[0], ks[1], np.abs(outs[ks[0]] - outs[ks[1]]).max()))
