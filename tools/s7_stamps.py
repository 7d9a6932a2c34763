#!/usr/bin/env python3
"""Where the waves of ResNet's streaming stem (csrc/stem7s_stream.hip) spend their cycles, per step phase.  Needs the stamped build:
    HSEFR_DEV=1 bash hse_facerec_tf_amd/csrc/build.sh && bash tools/build_ko.sh stem7s_stream.hip HSEFR_S7_STAMPS s7st 1
    HSEFR_LIB=libhsefr_s7st1.so python tools/s7_stamps.py
"""
import ctypes, os, sys
import numpy as np
os.environ.setdefault("HSEFR_LIB", "libhsefr_s7st1.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hse_facerec_tf_amd import _lib, ops, resnet50
rs = np.random.RandomState(1)
wp = ops.bf16_from_bits(resnet50.pack_stem_weight((rs.randn(7, 7, 3, 64) * 0.02).astype(np.float32)))
sc = torch.from_numpy(rs.uniform(0.5, 1.5, 64).astype(np.float32)).cuda(); sh = torch.from_numpy(rs.randn(64).astype(np.float32)).cuda()
x = torch.from_numpy(rs.uniform(-128, 128, (128, 224, 224, 3)).astype(np.float32)).cuda()
for _ in range(4): ops.stem7x7_pool_bf16(x, wp, sc, sh)
torch.cuda.synchronize()
buf = np.zeros((512 * 4, 10), np.uint64)
_lib.check(_lib.lib().hsefr_debug_read_stamps(8, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
r = buf.astype(np.float64); r = r[r[:, 9] > 0]
names = ["window rows: wait + convert + write", "next loads issued", "half 0: fragments + MFMAs", "half 0: epilogue", "half 1: fragments + MFMAs", "half 1: epilogue",
         "staging -> stores", "prologue + unit set-up (whole)"]
print("%d waves, lifetime %.0f cycles (min %.0f max %.0f), %.1f steps" % (len(r), r[:, 8].mean(), r[:, 8].min(), r[:, 8].max(), r[:, 9].mean()))
for i, nm in enumerate(names):
    print("   %-38s %5.1f %%  %7.0f cycles per step" % (nm, 100 * (r[:, i] / r[:, 8]).mean(), (r[:, i] / r[:, 9]).mean()))
