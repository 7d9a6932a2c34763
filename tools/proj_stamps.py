#!/usr/bin/env python3
"""Phase stamps of the increase layer + projected shortcut (csrc/conv1x1_bf16.hip PROJ) at ResNet-50's first three stage entries, batch 128:
    python tools/proj_stamps.py        (needs libhsefr_stamp.so: the development build with -DHSEFR_CD_STAMPS)"""
import ctypes, os, sys
import numpy as np
os.environ.setdefault("HSEFR_LIB", "libhsefr_stamp.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hse_facerec_tf_amd import _lib, ops
g = torch.Generator(device="cuda").manual_seed(0)
B = 128
for (oh, c, cout, c2, s2) in ((56, 64, 256, 64, 1), (28, 128, 512, 256, 2), (14, 256, 1024, 512, 2)):
    h2 = oh * s2
    x = (torch.rand((B, oh, oh, c), device="cuda", generator=g) * 2).to(torch.bfloat16)
    x2 = (torch.rand((B, h2, h2, c2), device="cuda", generator=g) * 2).to(torch.bfloat16)
    w = (torch.randn((cout, c), device="cuda", generator=g) / c ** 0.5).to(torch.bfloat16)
    w2 = (torch.randn((cout, c2), device="cuda", generator=g) / c2 ** 0.5).to(torch.bfloat16)
    sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    for _ in range(5):
        ops.conv1x1_proj_bf16(x, w, sc, sh, x2, w2, sc, sh, s2, 1)
    torch.cuda.synchronize()
    buf = np.zeros(512 * 4 * 8 - 2, np.uint64)
    _lib.check(_lib.lib().hsefr_debug_read_stamps(4, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
    b = np.concatenate([buf, [0, 0]]).astype(np.float64).reshape(512, 4, 8)
    rr = b.reshape(-1, 8); rr = rr[rr[:, 7] > 0]
    print("proj+inc %dx%d c%d+%d->%d: %d waves, lifetime %.0f cycles, %.1f steps -> %.0f cycles per step" % (oh, oh, c, c2, cout, len(rr), rr[:, 6].mean(), rr[:, 7].mean(), (rr[:, 6] / rr[:, 7]).mean()))
    for i, nm in enumerate(["loads issued + reads + MFMA", "wait for loads + LDS stage writes", "step barrier", "epilogue(s)", "barrier behind it"]):
        print("   %-34s %5.1f %%  %7.0f cycles per step" % (nm, 100 * (rr[:, i] / rr[:, 6]).mean(), (rr[:, i] / rr[:, 7]).mean()))
