#!/usr/bin/env python3
"""Where the waves of the pre-split GEMM (csrc/pwconv_ps.hip) spend their cycles.  Needs the stamped development build:
    HSEFR_DEV=1 HSEFR_EXTRA_FLAGS=-DHSEFR_PS_STAMPS bash hse_facerec_tf_amd/csrc/build.sh
"""
import ctypes
import os
import sys

import numpy as np

os.environ.setdefault("HSEFR_LIB", "libhsefr_dev.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hse_facerec_tf_amd import _lib, ops

B = 256
g = torch.Generator(device="cuda").manual_seed(0)
GRID = int(os.environ.get("PS_GRID", "0"))          # > 0: that many persistent workgroups, and M scaled down to keep 2 tiles each
_lib.check(_lib.lib().hsefr_debug_set(b"ps_grid", GRID))
for hw, k, n in ((12, 512, 512), (6, 1024, 1024), (24, 128, 256)):
    m = B * hw * hw
    if GRID:
        m = m * GRID // 256
    x = torch.rand((m, k), device="cuda", generator=g) * 6
    w = torch.randn((n, k), device="cuda", generator=g) / k ** 0.5
    sh = torch.randn((n,), device="cuda", generator=g)
    prep = ops.split_weights_device(w, x.device)
    xs = ops.split_rows_encode(x)
    DWE = os.environ.get("PS_DW") == "1" and hw in (12, 6)
    if DWE:          # the variant with the next depthwise in the epilogue (whole maps per tile)
        xs6 = xs.reshape(B * (GRID or 256) // 256 if GRID else B, hw, hw, k // 32, 2, 32)
        dw_w = torch.randn((3, 3, n), device="cuda", generator=g) / 3
        dsc = torch.rand((n,), device="cuda", generator=g) + 0.5
        dsh = torch.randn((n,), device="cuda", generator=g) * 0.3
    for _ in range(5):
        if DWE:
            ops.pwconv1x1_presplit_dw(xs6, None, sh, dw_w, dsc, dsh, prepared=prep)
        else:
            ops.pwconv1x1_presplit(xs, None, sh, prepared=prep)
    torch.cuda.synchronize()
    buf = np.zeros((256, 12, 8), np.uint64)
    _lib.check(_lib.lib().hsefr_debug_read_stamps(2, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
    b = buf.astype(np.float64)
    for role, sl, names in (("MFMA waves", slice(0, 8), ["ds_read + mfma issue", "step barrier", "epilogue (DW: the depthwise)", "tile barrier", "DW: parking", "DW: barrier waits"]),
                            ("loader waves", slice(8, 12), ["DMA issue", "vmcnt wait", "step barrier", "tile barrier / epilogue set-up", "DW: barrier waits", "DW: the depthwise"])):
        r = b[:, sl, :].reshape(-1, 8)
        r = r[r[:, 7] > 0]
        print("%dx%dx%d %s: %d waves, lifetime %.0f cycles (min %.0f max %.0f), %.1f steps -> %.0f cycles per step" %
              (m, k, n, role, len(r), r[:, 6].mean(), r[:, 6].min(), r[:, 6].max(), r[:, 7].mean(), (r[:, 6] / r[:, 7]).mean()))
        for i, nm in enumerate(names):
            print("   %-22s %5.1f %%  %7.0f cycles per step" % (nm, 100 * (r[:, i] / r[:, 6]).mean(), (r[:, i] / r[:, 7]).mean()))
