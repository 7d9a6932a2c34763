"""Forward time by batch size for plan variants (default | pwdw_fusion none | + block_fusion none): which plan should small batches run?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB, load_graph
from hse_facerec_tf_amd.lowering import lower_graph, OUT_FEATURES
from hse_facerec_tf_amd.engine import Engine
S = int(os.environ.get("KB_SIZE", 192))
g = load_graph(AGE_GENDER_PB, '')
variants = {"default": {}, "pwdw_none": {"pwdw_fusion": "none"}, "pwdw+block_none": {"pwdw_fusion": "none", "block_fusion": "none"},
            "presplit_none": {"presplit": "none"}}
engs = {}
for k, kw in variants.items():
    plan = lower_graph(g, "input_1:0", {OUT_FEATURES: "global_pooling/Mean:0"}, (S, S), {}, input_bound=256.0, **kw)
    engs[k] = Engine(plan, max_batch=64, device=0)
dev = torch.device("cuda:0")
for n in (1, 2, 4, 8, 16, 32, 64):
    x = (torch.rand((n, S, S, 3), device=dev) * 256 - 128).contiguous()
    ref = None
    line = "n=%2d " % n
    for k, e in engs.items():
        for _ in range(20): o = e.forward(x)["features"]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): o = e.forward(x)["features"]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200 * 1e3
        if ref is None: ref = o.clone()
        err = float((o - ref).abs().max() / ref.abs().max())
        line += " %s %.3f ms (%.1e)" % (k, dt, err)
    print(line)
