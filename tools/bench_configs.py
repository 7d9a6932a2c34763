#!/usr/bin/env python3
"""Throughput of the other BASELINE configs on one MI355X (parity-test cases, not the headline bench line):
    python tools/bench_configs.py resnet50      # config 3: ResNet-50, batch 128, 224x224x3, bf16 MFMA
    python tools/bench_configs.py agegender     # config 4: age/gender MobileNet multi-head, batch 512, 224x224x3
    python tools/bench_configs.py mobilenet192  # config 2 (the headline): MobileNet-192 embeddings, batch 256 -- for its per-LAYER table
    python tools/bench_configs.py mobilenet_f32 # config 2 with every product on the fp32 pipes (pw_math='f32')
Every run prints a per-LAYER table (the engine's HIP-event ring) and `forwards N`, the number of forwards it launched -- what
tools/make_profile_summary.py stores next to a PMC profile of the same command so that bench.py can turn bytes per launch into bytes per forward.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hse_facerec_tf_amd import lowering, resnet50
from hse_facerec_tf_amd.engine import Engine
from hse_facerec_tf_amd.graphdef import read_graph
from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB

STEPS, WARM = int(os.environ.get("BC_STEPS", "20")), 5


def run(eng, x, want, label, flops_per_img, bytes_per_img):
    for _ in range(WARM):
        out = eng.forward(x, want)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        out = eng.forward(x, want)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    n = x.shape[0]
    print("%s: %.3f ms/step  %.0f faces/s  (%.1f TFLOP/s, %.0f GB/s algorithmic)" %
          (label, dt * 1e3, n / dt, flops_per_img * n / dt / 1e12, bytes_per_img * n / dt / 1e9))
    eng.set_profiling(STEPS)
    for _ in range(STEPS):
        eng.forward(x, want)
    per = np.mean([eng.op_times_ms(s) for s in range(STEPS)], axis=0)
    eng.set_profiling(0)
    print("forwards %d" % (WARM + 2 * STEPS))
    for i, L in enumerate(eng.plan.layers):
        oh, ow, co = L.out_shape
        fl = float(lowering.Plan.layer_flops(L)) * n
        print("  %2d kind %2d %-26s %-16s -> %-16s %8.1f us %7.1f TF" % (i, L.kind, L.name[:26], L.in_shape, L.out_shape, per[i] * 1e3,
                                                                          fl / (per[i] * 1e-3) / 1e12 if per[i] > 0 else 0))
    return out


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    rs = np.random.RandomState(123)
    if what == "resnet50f32":
        B = int(os.environ.get("BC_BATCH", "32"))
        plan = resnet50.build_plan(resnet50.synthetic_weights(123), (224, 224), "caffe", dtype="f32")
        eng = Engine(plan, max_batch=B)
        x = torch.from_numpy(rs.uniform(-128, 128, (B, 224, 224, 3)).astype(np.float32)).cuda()
        run(eng, x, (0,), "ResNet-50 fp32-grade batch %d" % B, resnet50.flops_per_image(plan), 2 * resnet50.activation_bytes_per_image(plan))
    elif what == "resnet50":
        B = int(os.environ.get("BC_BATCH", "128"))
        plan = resnet50.build_plan(resnet50.synthetic_weights(123), (224, 224), "caffe")
        eng = Engine(plan, max_batch=B)
        x = torch.from_numpy(rs.uniform(-128, 128, (B, 224, 224, 3)).astype(np.float32)).cuda()
        run(eng, x, (0,), "ResNet-50 bf16 batch %d" % B, resnet50.flops_per_image(plan), resnet50.activation_bytes_per_image(plan))
    elif what in ("mobilenet192", "mobilenet_f32"):
        B = int(os.environ.get("BC_BATCH", "256"))
        f32 = what == "mobilenet_f32"
        plan = lowering.lower_graph(read_graph(AGE_GENDER_PB), "input_1:0", {0: "global_pooling/Mean:0"}, (192, 192),
                                    **({"pw_math": "f32"} if f32 else {"input_bound": 256.0}))
        eng = Engine(plan, max_batch=B)
        x = torch.from_numpy(rs.uniform(-128, 128, (B, 192, 192, 3)).astype(np.float32)).cuda()
        run(eng, x, (0,), "MobileNet-192 batch %d%s" % (B, " (pw_math='f32')" if f32 else ""), plan.flops_per_image(), plan.bytes_per_image())
    else:
        B = int(os.environ.get("BC_BATCH", "512"))
        plan = lowering.lower_graph(read_graph(AGE_GENDER_PB), "input_1:0",
                                    {0: "global_pooling/Mean:0", 1: "age_pred/Softmax:0", 2: "gender_pred/Sigmoid:0"}, input_bound=256.0)
        eng = Engine(plan, max_batch=B)
        x = torch.from_numpy(rs.uniform(-128, 128, (B, 224, 224, 3)).astype(np.float32)).cuda()
        run(eng, x, (0, 1, 2), "age/gender MobileNet-224 batch %d (3 outputs)" % B, plan.flops_per_image(), plan.bytes_per_image())
