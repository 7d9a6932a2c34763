#!/usr/bin/env python3
"""Does running two batches concurrently (two engines, two HIP streams) raise throughput?  One MI355X, MobileNet-192, batch 256 each.
    python tools/two_stream_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hse_facerec_tf_amd import lowering
from hse_facerec_tf_amd.engine import Engine
from hse_facerec_tf_amd.graphdef import read_graph
from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB

B, STEPS = 256, 60
plan = lowering.lower_graph(read_graph(AGE_GENDER_PB), "input_1:0", {0: "global_pooling/Mean:0"}, (192, 192))
x = [torch.rand((B, 192, 192, 3), device="cuda") * 255 - 128 for _ in range(2)]
engs = [Engine(plan, max_batch=B) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]


def run(nstreams):
    for _ in range(10):
        for i in range(nstreams):
            with torch.cuda.stream(streams[i]):
                engs[i].forward(x[i], (0,))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for s in range(STEPS):
        i = s % nstreams
        with torch.cuda.stream(streams[i]):
            engs[i].forward(x[i], (0,))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("%d stream(s): %.3f ms per batch, %.0f faces/s" % (nstreams, dt / STEPS * 1e3, B * STEPS / dt))


run(1)
run(2)
run(1)
