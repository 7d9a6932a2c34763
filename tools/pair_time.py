#!/usr/bin/env python3
"""Timing of csrc/conv1x1_pair_bf16.hip at ResNet-50's stage-2 shape (batch 128, 56 x 56, 64 -> 256 -> 64) against the two launches it
replaces, back to back on one box, with the development library's knock-outs (hsefr_debug_set "pair_ablate": 1 = no residual loads,
2 = no y1 stores, 4 = no y2 stores, 8 = no second product; results are then WRONG -- timing only).
usage: HSEFR_LIB=libhsefr_dev.so python tools/pair_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hse_facerec_tf_amd import _lib, ops

n, hw = int(os.environ.get("PT_BATCH", "128")), 56
g = torch.Generator(device="cuda").manual_seed(1)
x = (torch.rand((n, hw, hw, 64), device="cuda", generator=g) * 2).to(torch.bfloat16)
x2 = (torch.rand((n, hw, hw, 64), device="cuda", generator=g) * 2).to(torch.bfloat16)
r = (torch.rand((n, hw, hw, 256), device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
w1 = (torch.randn((256, 64), device="cuda", generator=g) / 8).to(torch.bfloat16)
wp = (torch.randn((256, 64), device="cuda", generator=g) / 8).to(torch.bfloat16)
w2 = (torch.randn((64, 256), device="cuda", generator=g) / 16).to(torch.bfloat16)
s1, b1 = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda") * 0.1
sp, bp = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda") * 0.1
s2, b2 = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def two(proj):
    y1 = ops.conv1x1_proj_bf16(x, w1, s1, b1, x2, wp, sp, bp, 1, 1) if proj else ops.conv_bf16(x, w1, s1, b1, 1, 1, 1, 0, r, 1)
    return ops.conv_bf16(y1, w2, s2, b2, 1, 1, 1, 0, None, 1)


def pair(proj):
    if proj:
        return ops.conv1x1_pair_bf16(x, w1, s1, b1, w2, s2, b2, x2=x2, wp_packed=wp, scale_p=sp, shift_p=bp)
    return ops.conv1x1_pair_bf16(x, w1, s1, b1, w2, s2, b2, res=r)


dev = hasattr(_lib.lib(), "hsefr_debug_set")
for proj in (False, True):
    print("%s: two launches %.1f us, pair %.1f us" % ("PROJ" if proj else "residual", timed(lambda: two(proj)), timed(lambda: pair(proj))))
    if dev:
        for abl in (1, 2, 4, 8, 3, 6, 7, 15):
            _lib.check(_lib.lib().hsefr_debug_set(b"pair_ablate", abl))
            print("   ablate %2d: %.1f us" % (abl, timed(lambda: pair(proj))))
        _lib.check(_lib.lib().hsefr_debug_set(b"pair_ablate", 0))
