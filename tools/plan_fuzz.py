"""Differential fuzz of the lowering / kernels: random input sizes and batches, the default plan against the least fused plan
(and the uint8 entry against the float entry).  Any disagreement beyond fp32 round-off is a bug in a fused kernel's edge handling.
usage: python tools/plan_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB, load_graph
from hse_facerec_tf_amd.lowering import lower_graph
from hse_facerec_tf_amd.engine import Engine

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = load_graph(AGE_GENDER_PB, '')
outs = {0: "global_pooling/Mean:0", 1: "age_pred/Softmax:0", 2: "gender_pred/Sigmoid:0"}
MEAN = (103.939, 116.779, 123.68)
worst = 0.0
for case in range(cases):
    h = int(rs.randint(32, 260)); w = int(rs.randint(32, 260))
    if rs.rand() < 0.6: h, w = h // 4 * 4, w // 4 * 4          # the streaming stem's shapes
    if rs.rand() < 0.3: w = h
    n = int(rs.choice([1, 2, 3, 5, 8, 17, 40]))
    try:
        base = lower_graph(g, "input_1:0", outs, (h, w), input_bound=256.0, u8_mean_bgr=MEAN)
        plain = lower_graph(g, "input_1:0", outs, (h, w), stem_fusion="none", block_fusion="none", presplit="none", pw_math="f32")
    except Exception as e:
        print("case %d %dx%d: lowering refused: %r" % (case, h, w, e)); continue
    ea, eb = Engine(base, max_batch=n), Engine(plain, max_batch=n)
    gen = torch.Generator(device="cuda"); gen.manual_seed(case)
    u8 = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=gen)
    x = (u8.flip(-1).double() - torch.tensor(MEAN, dtype=torch.float64, device="cuda")).float().contiguous()
    ra, rb = ea.forward(x, (0, 1, 2)), eb.forward(x, (0, 1, 2))
    line = "case %2d  %3dx%3d n=%2d kinds %s" % (case, h, w, n, [L.kind for L in base.layers][:3])
    for k in ("features", "age_probs", "gender"):
        d = float((ra[k] - rb[k]).abs().max() / (rb[k].abs().max() + 1e-30)); worst = max(worst, d)
        line += "  %s %.1e" % (k[:4], d)
        assert d < 2e-5, line
    if ea.accepts_u8:
        ru = ea.forward_u8(u8, (0,))["features"]
        d = float((ru - ra["features"]).abs().max() / (ra["features"].abs().max() + 1e-30)); worst = max(worst, d)
        line += "  u8 %.1e" % d
        assert d < 2e-5, line
    assert not ea.input_overflow()
    print(line)
    ea.close(); eb.close()
print("worst relative difference %.2e over %d cases" % (worst, cases))
