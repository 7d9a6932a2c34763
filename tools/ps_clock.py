#!/usr/bin/env python3
"""The clock the chip holds under the pre-split GEMM (csrc/pwconv_ps.hip) run back to back: wave lifetimes in s_memtime ticks (= shader
cycles) against the launch's duration by events.  Needs a development build with the lifetime stamps only:
    HSEFR_DEV=1 HSEFR_EXTRA_FLAGS=-DHSEFR_PS_STAMPS=2 bash hse_facerec_tf_amd/csrc/build.sh
(DESIGN.md lesson 56; results of the round-5 ablations: profiles/r05_ps_ablation.txt)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hse_facerec_tf_amd import _lib, ops
B = 256
g = torch.Generator(device="cuda").manual_seed(0)
for hw, k, n in ((12, 512, 512),):
    m = B * hw * hw
    x = torch.rand((m, k), device="cuda", generator=g) * 6
    w = torch.randn((n, k), device="cuda", generator=g) / k ** 0.5
    sh = torch.randn((n,), device="cuda", generator=g)
    prep = ops.split_weights_device(w, x.device)
    xs = ops.split_rows_encode(x)
    fn = lambda: ops.pwconv1x1_presplit(xs, None, sh, prepared=prep)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record()
    for i in range(40):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(40))
    us = ts[20] * 1e3
    buf = np.zeros((256, 12, 8), np.uint64)
    _lib.check(_lib.lib().hsefr_debug_read_stamps(2, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
    life = buf[:, :, 6].astype(np.float64)
    print(os.environ.get("HSEFR_LIB"), "median %.1f us, wave lifetime mean %.0f max %.0f ticks -> %.2f ticks/ns" % (us, life.mean(), life.max(), life.max() / us / 1e3))
