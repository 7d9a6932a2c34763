#!/usr/bin/env python3
"""Times the fused-stem generations on the BASELINE shape (batch 256, 192x192x3) in one process, product library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hse_facerec_tf_amd import ops

B = int(os.environ.get("KB_BATCH", "256"))


def timeit(fn, iters=20, warm=3):
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


g = torch.Generator(device="cuda").manual_seed(0)
for hw in (192, 224):
    xs = [((torch.rand((B, hw, hw, 3), device="cuda", generator=g) - 0.5) * 255).contiguous() for _ in range(3)]
    x8 = [(x + 128).clamp(0, 255).to(torch.uint8).contiguous() for x in xs]
    cw = torch.randn((3, 3, 3, 32), device="cuda", generator=g) * 0.02
    csh = torch.randn((32,), device="cuda", generator=g)
    wd = torch.randn((3, 3, 32), device="cuda", generator=g) / 3
    dsc = torch.rand((32,), device="cuda", generator=g) + 0.5
    dsh = torch.randn((32,), device="cuda", generator=g) * 0.3
    w = torch.randn((64, 32), device="cuda", generator=g) / 32 ** 0.5
    sh = torch.randn((64,), device="cuda", generator=g)
    wd2 = torch.randn((3, 3, 64), device="cuda", generator=g) / 3
    d2sc = torch.rand((64,), device="cuda", generator=g) + 0.5
    d2sh = torch.randn((64,), device="cuda", generator=g) * 0.3
    prep = ops.split_weights_device(w, xs[0].device)
    k = [0]

    def nx(lst):
        k[0] += 1
        return lst[k[0] % 3]
    t2 = timeit(lambda: ops.stem2_fused(nx(xs), cw, csh, wd, dsc, dsh, None, sh, wd2, d2sc, d2sh, prepared=prep))
    t3 = timeit(lambda: ops.stem3_fused(nx(xs), cw, csh, wd, dsc, dsh, None, sh, wd2, d2sc, d2sh, prepared=prep))
    t4 = timeit(lambda: ops.stem4_fused(nx(xs), cw, csh, wd, dsc, dsh, None, sh, wd2, d2sc, d2sh, prepared=prep))
    t8 = timeit(lambda: ops.stem4_fused(nx(x8), cw, csh, wd, dsc, dsh, None, sh, wd2, d2sc, d2sh, prepared=prep, u8_mean_bgr=(103.939, 116.779, 123.68)))
    print("stem %d batch %d: stem2 %.1f (min %.1f)  stem3 %.1f (%.1f)  stem4 %.1f (%.1f)  stem4-u8 %.1f (%.1f) us  [ops wrappers include ~20 us of host-side packing]"
          % (hw, B, t2[0], t2[1], t3[0], t3[1], t4[0], t4[1], t8[0], t8[1]))
