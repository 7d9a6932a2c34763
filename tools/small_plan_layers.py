import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB, load_graph
from hse_facerec_tf_amd.lowering import lower_graph, OUT_FEATURES
from hse_facerec_tf_amd.engine import Engine
g = load_graph(AGE_GENDER_PB, '')
plan = lower_graph(g, "input_1:0", {OUT_FEATURES: "global_pooling/Mean:0"}, (192, 192), {}, input_bound=256.0, presplit="none")
e = Engine(plan, max_batch=8, device=0)
for n in (1, 8):
    x = (torch.rand((n, 192, 192, 3), device="cuda") * 256 - 128).contiguous()
    e.set_profiling(64)
    for _ in range(64): e.forward(x)
    torch.cuda.synchronize()
    import numpy as np
    t = np.median(np.array([e.op_times_ms(i) for i in range(8, 60)]), axis=0) * 1e3
    print("n=%d total %.1f us over %d ops" % (n, t.sum(), len(t)))
    for L, us in zip(plan.layers, t): print("   kind %2d %-40s out %-16s %6.1f us" % (L.kind, L.name[:40], L.out_shape, us))
    e.set_profiling(0)
