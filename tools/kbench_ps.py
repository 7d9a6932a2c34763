#!/usr/bin/env python3
"""Pointwise GEMMs of MobileNet-192 @ batch 256 (and the depthwise layers in front of them): the register-staged split-f16
kernel on fp32 activations vs the LDS-DMA kernel on pre-split activations, interleaved in ONE process.
    python tools/kbench_ps.py            (product library; no tuning knobs needed)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hse_facerec_tf_amd import ops

B = int(os.environ.get("KB_BATCH", "256"))
ITERS = int(os.environ.get("KB_ITERS", "30"))
PW = [(48, 64, 128), (24, 128, 256), (12, 256, 512), (12, 512, 512), (6, 512, 1024), (6, 1024, 1024)]
DW = [(48, 128, 2), (24, 256, 2), (12, 512, 1), (12, 512, 2), (6, 1024, 1)]


def timeit(fn, iters=ITERS, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    print("%-22s %12s %12s   %8s %8s" % ("M x K x N", "f16s fp32-in", "presplit", "TF(f16)", "GB/s"))
    for hw, k, n in PW:
        m = B * hw * hw
        x = torch.rand((m, k), device="cuda", generator=g) * 6
        w = torch.randn((n, k), device="cuda", generator=g) / k ** 0.5
        sh = torch.randn((n,), device="cuda", generator=g)
        prep = ops.split_weights_device(w, x.device)
        xs = ops.split_rows_encode(x)
        junk = torch.empty(64 * 1024 * 1024, device="cuda")       # 256 MB: evicts the Infinity Cache between launches

        def cold(fn):
            def f():
                junk.zero_()
                return fn()
            return f
        t_old, _ = timeit(lambda: ops.pwconv1x1_f16split(x, None, sh, prepared=prep))
        t_new, mn = timeit(lambda: ops.pwconv1x1_presplit(xs, None, sh, prepared=prep))
        flops = 2.0 * m * k * n * 3
        nbytes = 4.0 * m * (k + n) + 4.0 * k * n
        print("%-22s %9.1f us %9.1f us   %8.0f %8.0f   (min %.1f)" % ("%d x %d x %d" % (m, k, n), t_old, t_new, flops / t_new / 1e6,
                                                                        nbytes / t_new / 1e3, mn))
    print("\ndepthwise: fp32 store vs split-row store")
    for hw, c, s in DW:
        x = torch.rand((B, hw, hw, c), device="cuda", generator=g) * 6
        kd = torch.randn((3, 3, c), device="cuda", generator=g) / 3
        sc = torch.rand((c,), device="cuda", generator=g) + 0.5
        sh = torch.randn((c,), device="cuda", generator=g) * 0.3
        t0, _ = timeit(lambda: ops.dwconv3x3(x, kd, sc, sh, s))
        t1, _ = timeit(lambda: ops.dwconv3x3_split(x, kd, sc, sh, s))
        oh = (hw + s - 1) // s
        nb = 4.0 * B * c * (hw * hw + oh * oh)
        print("%3dx%-3d c%-5d s%d   %8.1f us %8.1f us   %6.0f GB/s" % (hw, hw, c, s, t0, t1, nb / t1 / 1e3))


if __name__ == "__main__":
    main()
