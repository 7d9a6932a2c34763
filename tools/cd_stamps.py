#!/usr/bin/env python3
"""Where the waves of the bf16 LDS-DMA convolution (csrc/conv_dma_bf16.hip) spend their cycles.  Needs the stamped development build:
    HSEFR_DEV=1 HSEFR_EXTRA_FLAGS=-DHSEFR_CD_STAMPS bash hse_facerec_tf_amd/csrc/build.sh      (rm -rf csrc/build_dev first)
    python tools/cd_stamps.py [layer ...]      (layer names of tools/kbench_conv.py)
"""
import ctypes
import os
import sys

import numpy as np

os.environ.setdefault("HSEFR_LIB", "libhsefr_dev.so")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hse_facerec_tf_amd import _lib, ops
from kbench_conv import LAYERS, B

g = torch.Generator(device="cuda").manual_seed(0)
WIN = os.environ.get("CD_WIN", "0") == "1"       # CD_WIN=1: the window 3x3 kernel (csrc/conv3x3_win_bf16.hip) instead
W2 = os.environ.get("CD_W2", "0") == "1"         # CD_W2=1: the four-wave window kernel (csrc/conv3x3_w2_bf16.hip)
W4 = os.environ.get("CD_W4", "0") == "1"         # CD_W4=1: the four-wave 1x1 GEMM (csrc/conv1x1_w4_bf16.hip)
C11 = os.environ.get("CD_C11", "0") == "1"       # CD_C11=1: the register-staged persistent 1x1 GEMM (csrc/conv1x1_bf16.hip)
_lib.check(_lib.lib().hsefr_debug_set(b"cd_off", 1 if C11 else 2))
_lib.check(_lib.lib().hsefr_debug_set(b"w3_off", 2 if WIN else 1))
_lib.check(_lib.lib().hsefr_debug_set(b"w2_off", 2 if W2 else 1))
_lib.check(_lib.lib().hsefr_debug_set(b"w4_off", 2 if W4 else 1))
RB = int(os.environ.get("CD_RB", "0"))
_lib.check(_lib.lib().hsefr_debug_set(b"cd_rb", RB))
for name in sys.argv[1:] or ["c4_3x3", "c4_red", "c4_inc"]:
    hw, c, cout, k, s, res = LAYERS[name]
    x = (torch.rand((B, hw, hw, c), device="cuda", generator=g) * 2).to(torch.bfloat16)
    w = (torch.randn((cout, k * k * c), device="cuda", generator=g) / (k * k * c) ** 0.5).to(torch.bfloat16)
    sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    oh = (hw + 2 * (k // 2) - k) // s + 1
    r = (torch.rand((B, oh, oh, cout), device="cuda", generator=g)).to(torch.bfloat16) if res else None
    for _ in range(5):
        ops.conv_bf16(x, w, sc, sh, k, k, stride=s, pad=k // 2, res=r)
    torch.cuda.synchronize()
    if C11:
        buf = np.zeros(512 * 4 * 8 - 2, np.uint64)
        _lib.check(_lib.lib().hsefr_debug_read_stamps(4, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
        b = np.concatenate([buf, [0, 0]]).astype(np.float64).reshape(512, 4, 8)
        roles = (("all four waves", slice(0, 4), ["loads issued + reads + MFMA", "wait for loads + LDS stage writes", "step barrier", "epilogue", "barrier behind it"]),)
    elif W2 or W4:
        buf = np.zeros(256 * 8 * 8 - (1 if W4 else 0), np.uint64)
        _lib.check(_lib.lib().hsefr_debug_read_stamps(5 if W4 else 6, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
        b = np.concatenate([buf, [0] * (1 if W4 else 0)]).astype(np.float64).reshape(256, 8, 8)
        roles = (("MFMA waves", slice(0, 4), ["ds_read + mfma issue", "step barrier", "epilogue"]),
                 ("loader waves", slice(4, 8), ["DMA issue", "vmcnt wait", "step barrier"]))
    else:
        buf = np.zeros(256 * 12 * 8 - (2 if WIN else 1), np.uint64)
        _lib.check(_lib.lib().hsefr_debug_read_stamps(7 if WIN else 3, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
        b = np.concatenate([buf, [0] * (2 if WIN else 1)]).astype(np.float64).reshape(256, 12, 8)
        roles = (("MFMA waves", slice(0, 8), ["ds_read + mfma issue", "step barrier", "epilogue", "tile barrier"]),
                 ("loader waves", slice(8, 12), ["DMA issue", "vmcnt wait", "step barrier", "tile barrier"]))
    for role, sl, names in roles:
        rr = b[:, sl, :].reshape(-1, 8)
        rr = rr[rr[:, 7] > 0]
        print("%s %s: %d waves, lifetime %.0f cycles (min %.0f max %.0f), %.1f steps -> %.0f cycles per step" %
              (name, role, len(rr), rr[:, 6].mean(), rr[:, 6].min(), rr[:, 6].max(), rr[:, 7].mean(), (rr[:, 6] / rr[:, 7]).mean()))
        for i, nm in enumerate(names):
            print("   %-22s %5.1f %%  %7.0f cycles per step" % (nm, 100 * (rr[:, i] / rr[:, 6]).mean(), (rr[:, i] / rr[:, 7]).mean()))
