#!/usr/bin/env python3
"""Is the MobileNet-192 forward pass (batch 256) power-limited?  (1) pass time by events, back to back and with idle gaps between passes;
(2) rocm-smi power / clock samples during six seconds of back-to-back passes.  Product library.  DESIGN.md lesson 56."""
import os, sys, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB, TensorFlowInference
B, S = 256, 192
tfi = TensorFlowInference(AGE_GENDER_PB, input_tensor="input_1:0", output_tensor="global_pooling/Mean:0", convert2BGR=True,
                          imageNetUtilsMean=True, input_size=(S, S), max_batch=B, device=0)
eng = tfi.engine
gen = torch.Generator(device="cuda").manual_seed(123)
xs = [(torch.rand((B, S, S, 3), device="cuda", generator=gen) * 256.0 - 128.0).contiguous() for _ in range(4)]
def passes(n, gap_ms):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        ev[i][0].record()
        eng.forward(xs[i & 3])
        ev[i][1].record()
        if gap_ms:
            torch.cuda.synchronize()
            time.sleep(gap_ms * 1e-3)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]
for _ in range(200): eng.forward(xs[0])
torch.cuda.synchronize()
for gap in (0, 0, 1, 3, 10, 0):
    n = 2000 if gap == 0 else 300
    med, mn = passes(n, gap)
    print("gap %2d ms: pass median %.4f ms  min %.4f ms" % (gap, med, mn), flush=True)
# smi samples under sustained load
stop = False
samples = []
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
            samples.append([l.strip() for l in o.splitlines() if any(k in l for k in ("Power", "sclk", "mclk", "junction", "edge"))])
        except Exception as e:
            samples.append([repr(e)])
th = threading.Thread(target=sampler); th.start()
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(100): eng.forward(xs[0])
    torch.cuda.synchronize()
stop = True; th.join()
for s in samples[:2] + samples[-3:]:
    print(s)
