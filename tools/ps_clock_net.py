#!/usr/bin/env python3
"""The clock the chip holds under the LAST pre-split GEMM of a MobileNet-192 forward pass (batch 256) inside back-to-back passes: its wave
lifetimes (s_memtime ticks = shader cycles, -DHSEFR_PS_STAMPS=2 development build as for tools/ps_clock.py) against its duration by the
engine's op events.  DESIGN.md lesson 56."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hse_facerec_tf_amd import _lib
from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB, TensorFlowInference
B, S = 256, 192
tfi = TensorFlowInference(AGE_GENDER_PB, input_tensor="input_1:0", output_tensor="global_pooling/Mean:0", convert2BGR=True,
                          imageNetUtilsMean=True, input_size=(S, S), max_batch=B, device=0)
eng = tfi.engine
gen = torch.Generator(device="cuda").manual_seed(123)
x = (torch.rand((B, S, S, 3), device="cuda", generator=gen) * 256.0 - 128.0).contiguous()
for _ in range(100): eng.forward(x)
torch.cuda.synchronize()
N = 20
eng.set_profiling(N)
for _ in range(N): eng.forward(x)
per = np.mean([eng.op_times_ms(s) for s in range(N)], axis=0)
eng.set_profiling(0)
for _ in range(50): eng.forward(x)          # (unprofiled forwards: the stamps below are from a back-to-back pass)
torch.cuda.synchronize()
buf = np.zeros((256, 12, 8), np.uint64)
_lib.check(_lib.lib().hsefr_debug_read_stamps(2, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
life = buf[:, :, 6].astype(np.float64)
last = len(eng.plan.layers) - 1
print("last pwconv_ps launch of the pass (layer %d, %s): %.1f us with op events; wave lifetime max %.0f ticks -> %.2f ticks/ns" %
      (last, eng.plan.layers[last].name, per[last] * 1e3, life.max(), life.max() / (per[last] * 1e6)))
