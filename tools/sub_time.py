#!/usr/bin/env python3
"""Costing of the stride-2 forms of a ResNet stage's last block (DESIGN.md lesson 72): the 3x3 layer at stride 2 and the increase layer on the
compact map against the full-size layers they replace, batch 128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hse_facerec_tf_amd import ops, resnet50
g = torch.Generator(device="cuda").manual_seed(0)
B = 128
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for hw, c in ((56, 64), (28, 128), (14, 256)):
    x = (torch.rand((B, hw, hw, c), device="cuda", generator=g) * 2).to(torch.bfloat16)
    w3 = (torch.randn((c, 9 * c), device="cuda", generator=g) / (9 * c) ** 0.5).to(torch.bfloat16)
    sc, sh = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    t1 = t(lambda: ops.conv_bf16(x, w3, sc, sh, 3, 3, 1, 1))
    t2 = t(lambda: ops.conv_bf16(x, w3, sc, sh, 3, 3, 2, 1))
    co = 4 * c
    wi = (torch.randn((co, c), device="cuda", generator=g) / c ** 0.5).to(torch.bfloat16)
    sco, sho = torch.ones(co, device="cuda"), torch.zeros(co, device="cuda")
    rf = (torch.rand((B, hw, hw, co), device="cuda", generator=g)).to(torch.bfloat16)
    xh = x[:, ::2, ::2, :].contiguous(); rh = rf[:, ::2, ::2, :].contiguous()
    t3 = t(lambda: ops.conv_bf16(x, wi, sco, sho, 1, 1, res=rf))
    t4 = t(lambda: ops.conv_bf16(xh, wi, sco, sho, 1, 1, res=rh))
    t5 = t(lambda: rf[:, ::2, ::2, :].contiguous())
    print("%dx%d c%d: 3x3 s1 %.1f us, s2 %.1f us; increase full %.1f us, compact %.1f us; torch gather of the residual %.1f us" % (hw, hw, c, t1, t2, t3, t4, t5))
