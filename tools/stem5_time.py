#!/usr/bin/env python3
"""Kernel-only timing of the stem generations on the BASELINE shape (batch 256, 192x192x3; KB_HW / KB_BATCH override):
every operand is prepared once, the timed loop calls the C entry points directly (no host-side packing inside it).
    HSEFR_LIB=libhsefr_dev.so KB_S5_GRID=768 KB_S5_SEGS=0 python tools/stem5_time.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hse_facerec_tf_amd import _lib, lowering, ops

B = int(os.environ.get("KB_BATCH", "256"))
HW = int(os.environ.get("KB_HW", "192"))
lib = _lib.lib()
for key, env in ((b"stem5_grid", "KB_S5_GRID"), (b"stem5_segs", "KB_S5_SEGS"), (b"stem4_grid", "KB_STEM4_GRID")):
    if env in os.environ:
        _lib.check(lib.hsefr_debug_set(key, int(os.environ[env])))
g = torch.Generator(device="cuda").manual_seed(0)
xs = [((torch.rand((B, HW, HW, 3), device="cuda", generator=g) - 0.5) * 255).contiguous() for _ in range(3)]
x8 = [(x + 128).clamp(0, 255).to(torch.uint8).contiguous() for x in xs]
cw = (torch.randn((3, 3, 3, 32), device="cuda", generator=g) * 0.02).cpu().numpy()
csh = torch.randn((32,), device="cuda", generator=g)
wd = (torch.randn((3, 3, 32), device="cuda", generator=g) / 3).contiguous()
dsc = torch.rand((32,), device="cuda", generator=g) + 0.5
dsh = torch.randn((32,), device="cuda", generator=g) * 0.3
w = torch.randn((64, 32), device="cuda", generator=g) / 32 ** 0.5
sh = torch.randn((64,), device="cuda", generator=g)
wd2 = (torch.randn((3, 3, 64), device="cuda", generator=g) / 3).contiguous()
d2sc = torch.rand((64,), device="cuda", generator=g) + 0.5
d2sh = torch.randn((64,), device="cuda", generator=g) * 0.3
d_img, d_ds = ops.split_weights_device(w, xs[0].device, 12)
img4, ds4 = lowering.stem4_conv_image(cw, 7)
d_c4, d_cds = torch.from_numpy(img4.view(np.int16)).cuda(), torch.from_numpy(ds4).cuda()
img8, ds8 = lowering.stem4_conv_image(cw, 0, reverse_channels=True)
d_c8, d_cds8 = torch.from_numpy(img8.view(np.int16)).cuda(), torch.from_numpy(ds8).cuda()
d_sh8 = torch.from_numpy(lowering.stem4_u8_shifts(cw, csh.cpu().numpy(), (103.939, 116.779, 123.68))).cuda()
y = torch.empty((B, HW // 4, HW // 4, 64), device="cuda")
st = _lib.current_stream_ptr()


def call(entry, u8, k):
    x = (x8 if u8 else xs)[k % 3]
    _lib.check(getattr(lib, entry)(x.data_ptr(), 1 if u8 else 0, (d_c8 if u8 else d_c4).data_ptr(), (d_cds8 if u8 else d_cds).data_ptr(),
                                   (d_sh8 if u8 else csh).data_ptr(), wd.data_ptr(), dsc.data_ptr(), dsh.data_ptr(), d_img.data_ptr(), d_ds.data_ptr(),
                                   sh.data_ptr(), wd2.data_ptr(), d2sc.data_ptr(), d2sh.data_ptr(), y.data_ptr(), None, B, HW, HW, 0 if u8 else 7, 12,
                                   ops.ACT_RELU6, st), entry)


def timeit(entry, u8, iters=30, warm=5):
    for i in range(warm):
        call(entry, u8, i)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        call(entry, u8, i)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


out = []
for entry in os.environ.get("KB_ENTRIES", "hsefr_stem4_fused,hsefr_stem5_stream").split(","):
    for u8 in (False, True):
        med, mn = timeit(entry, u8)
        out.append("%s%s %.1f (min %.1f)" % (entry.replace("hsefr_", ""), "-u8" if u8 else "", med, mn))
print("stem %dx%d batch %d [us]: %s" % (HW, HW, B, "  ".join(out)))
