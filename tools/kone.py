#!/usr/bin/env python3
"""One kernel, many launches (for rocprofv3 --pmc passes): python tools/kone.py pws 12 512 512 [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hse_facerec_tf_amd import ops

kind, hw, k, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 30
B = int(os.environ.get("KB_BATCH", "256"))
g = torch.Generator(device="cuda").manual_seed(0)
m = B * hw * hw
x = torch.rand((m, k), device="cuda", generator=g) * 6
w = torch.randn((n, k), device="cuda", generator=g) / k ** 0.5
sh = torch.randn((n,), device="cuda", generator=g)
if kind == "pws":
    prep = ops.split_weights_device(w, x.device)
    fn = lambda: ops.pwconv1x1_f16split(x, None, sh, prepared=prep)
elif kind == "pw":
    fn = lambda: ops.pwconv1x1(x, w, sh)
else:
    raise SystemExit("unknown kind")
for _ in range(iters):
    fn()
torch.cuda.synchronize()
