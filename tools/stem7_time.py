#!/usr/bin/env python3
"""Kernel-only timing of ResNet's fused stem (hsefr_stem7x7_pool_bf16) at batch 128 x 224 x 224 for the loaded library (HSEFR_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hse_facerec_tf_amd import ops, resnet50
rs = np.random.RandomState(1)
wp = ops.bf16_from_bits(resnet50.pack_stem_weight((rs.randn(7, 7, 3, 64) * 0.02).astype(np.float32)))
sc = torch.from_numpy(rs.uniform(0.5, 1.5, 64).astype(np.float32)).cuda(); sh = torch.from_numpy(rs.randn(64).astype(np.float32)).cuda()
xs = [torch.from_numpy(rs.uniform(-128, 128, (128, 224, 224, 3)).astype(np.float32)).cuda() for _ in range(3)]
for _ in range(5): ops.stem7x7_pool_bf16(xs[0], wp, sc, sh)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(60): ops.stem7x7_pool_bf16(xs[i % 3], wp, sc, sh)
e1.record(); torch.cuda.synchronize()
print("%s: %.1f us" % (os.environ.get("HSEFR_LIB", "libhsefr.so"), e0.elapsed_time(e1) / 60 * 1e3))
