#!/usr/bin/env python3
"""Phase stamps of the register-staged 1x1 GEMM (csrc/conv1x1_bf16.hip) on ResNet-50's residual increase layers, batch 128:
    HSEFR_DEV=1 bash hse_facerec_tf_amd/csrc/build.sh && bash tools/build_ko.sh conv1x1_bf16.hip HSEFR_CD_STAMPS c11st 1
    HSEFR_LIB=libhsefr_c11st1.so python tools/c11_stamps.py"""
import ctypes, os, sys
import numpy as np
os.environ.setdefault("HSEFR_LIB", "libhsefr_c11st1.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hse_facerec_tf_amd import _lib, ops
g = torch.Generator(device="cuda").manual_seed(0)
B = 128
for (oh, c, cout) in ((56, 64, 256), (28, 128, 512), (14, 256, 1024)):
    x = (torch.rand((B, oh, oh, c), device="cuda", generator=g) * 2).to(torch.bfloat16)
    r = (torch.rand((B, oh, oh, cout), device="cuda", generator=g) * 2).to(torch.bfloat16)
    w = (torch.randn((cout, c), device="cuda", generator=g) / c ** 0.5).to(torch.bfloat16)
    sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    for _ in range(5):
        ops.conv_bf16(x, w, sc, sh, 1, 1, res=r, act=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv_bf16(x, w, sc, sh, 1, 1, res=r, act=1)
    e1.record(); torch.cuda.synchronize()
    buf = np.zeros(512 * 4 * 8 - 2, np.uint64)
    _lib.check(_lib.lib().hsefr_debug_read_stamps(4, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
    b = np.concatenate([buf, [0, 0]]).astype(np.float64).reshape(512, 4, 8)
    rr = b.reshape(-1, 8); rr = rr[rr[:, 7] > 0]
    print("increase %dx%d c%d->%d: %.1f us; %d waves, lifetime %.0f cycles, %.1f steps -> %.0f cycles per step" %
          (oh, oh, c, cout, e0.elapsed_time(e1) / 20 * 1e3, len(rr), rr[:, 6].mean(), rr[:, 7].mean(), (rr[:, 6] / rr[:, 7]).mean()))
    for i, nm in enumerate(["loads issued + reads + MFMA", "wait for loads + LDS stage writes", "step barrier", "epilogue(s)", "barrier behind it"]):
        print("   %-34s %5.1f %%  %7.0f cycles per step" % (nm, 100 * (rr[:, i] / rr[:, 6]).mean(), (rr[:, i] / rr[:, 7]).mean()))
