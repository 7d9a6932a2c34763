#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X: every layer shape of MobileNet-192 @ batch 256 through
the per-kernel C-ABI entry points, interleaved variants in ONE process (cdna guide rule 24).
    python tools/kbench.py [pw] [dw] [c3]
Needs the DEVELOPMENT build of the library (tuning knobs, calibration kernels):
    HSEFR_DEV=1 bash hse_facerec_tf_amd/csrc/build.sh      -> hse_facerec_tf_amd/libhsefr_dev.so (selected below)
"""
import os
import sys

import numpy as np

os.environ.setdefault("HSEFR_LIB", "libhsefr_dev.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hse_facerec_tf_amd import _lib, ops

B = int(os.environ.get("KB_BATCH", "256"))
PW = [(96, 32, 64), (48, 64, 128), (48, 128, 128), (24, 128, 256), (24, 256, 256), (12, 256, 512), (12, 512, 512),
      (6, 512, 1024), (6, 1024, 1024)]
DW = [(96, 32, 1), (96, 64, 2), (48, 128, 1), (48, 128, 2), (24, 256, 1), (24, 256, 2), (12, 512, 1), (12, 512, 2), (6, 1024, 1)]


ITERS = int(os.environ.get("KB_ITERS", "20"))


def timeit(fn, iters=None, warm=3):
    iters = iters or ITERS
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
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3     # median, min in us


def bench_pw():
    print("pointwise fp32-MFMA GEMM, batch %d: us (median) per tile config" % B)
    print("%-22s %10s %10s %10s %10s   best-TF  GB/s" % ("M x K x N", "auto", "128x128", "128x64", "64x64"))
    g = torch.Generator(device="cuda").manual_seed(0)
    for hw, k, n in PW:
        m = B * hw * hw
        x = torch.rand((m, k), device="cuda", generator=g) * 6
        w = torch.randn((n, k), device="cuda", generator=g) / k ** 0.5
        sh = torch.randn((n,), device="cuda", generator=g)
        res = []
        for tile in (-1, 0, 1, 2):
            if tile == 0 and n % 128:
                res.append(float("nan"))
                continue
            _lib.lib().hsefr_debug_set(b"pw_tile", tile)
            res.append(timeit(lambda: ops.pwconv1x1(x, w, sh))[0])
        _lib.lib().hsefr_debug_set(b"pw_tile", -1)
        best = np.nanmin(res)
        fl = 2.0 * m * k * n
        by = 4.0 * (m * k + m * n + k * n)
        print("%-22s %10.1f %10.1f %10.1f %10.1f   %6.1f  %6.0f" % ("%dx%dx%d" % (m, k, n), *res, fl / best / 1e6, by / best / 1e3))


def bench_dw():
    ths = [0, 4, 8, 12, 16, 24, 48]
    print("depthwise 3x3, batch %d: us (median) per strip height; 0 = heuristic" % B)
    print("%-18s" % "layer" + "".join("%9s" % ("th=%d" % t) for t in ths) + "   best GB/s")
    g = torch.Generator(device="cuda").manual_seed(0)
    for hw, c, s in DW:
        x = torch.rand((B, hw, hw, c), device="cuda", generator=g) * 6
        w = torch.randn((3, 3, c), device="cuda", generator=g)
        sc = torch.rand((c,), device="cuda", generator=g) + 0.5
        sh = torch.randn((c,), device="cuda", generator=g)
        oh = (hw + s - 1) // s
        res = []
        for th in ths:
            if th > oh:
                res.append(float("nan"))
                continue
            _lib.lib().hsefr_debug_set(b"dw_th", th)
            res.append(timeit(lambda: ops.dwconv3x3(x, w, sc, sh, s))[0])
        _lib.lib().hsefr_debug_set(b"dw_th", 0)
        by = 4.0 * B * c * (hw * hw + oh * oh)
        print("%3dx%-3d c=%-4d s%d " % (hw, hw, c, s) + "".join("%9.1f" % r for r in res) + "   %7.0f" % (by / np.nanmin(res) / 1e3))


def bench_c3():
    g = torch.Generator(device="cuda").manual_seed(0)
    for hw in (192, 224):
        x = (torch.rand((B, hw, hw, 3), device="cuda", generator=g) - 0.5) * 256
        w = torch.randn((3, 3, 3, 32), device="cuda", generator=g) * 0.05
        sh = torch.randn((32,), device="cuda", generator=g)
        by = 4.0 * B * (hw * hw * 3 + (hw // 2) ** 2 * 32)
        for impl, name in ((1, "valu"), (2, "mfma")):
            _lib.lib().hsefr_debug_set(b"c3_impl", impl)
            med, mn = timeit(lambda: ops.conv3x3_c3(x, w, sh, 2))
            print("conv1 %d %s: %8.1f us (min %8.1f)  %7.0f GB/s" % (hw, name, med, mn, by / med / 1e3))
        _lib.lib().hsefr_debug_set(b"c3_impl", 0)


def bench_copy():
    print("copy calibration, GB/s read+write (median): rows = variant (unroll, nt-load, nt-store, WG/CU), cols = MB")
    sizes = (64, 302, 604, 1208)
    bufs = []
    for mb in sizes:
        n = mb * 1000 * 1000 // 16 * 16
        a = torch.rand((n // 4,), device="cuda")
        bufs.append((n, a, torch.empty_like(a)))
    for grid_bits, wg in ((1, 4), (0, 8), (2, 16), (3, 32)):
        for nt in (0, 2, 3):
            for ub, u in ((0, 1), (2, 4), (3, 8)):
                v = ub | (nt << 2) | (grid_bits << 4)
                _lib.lib().hsefr_debug_set(b"copy_variant", v)
                row = []
                for n, a, b in bufs:
                    med, mn = timeit(lambda: _lib.check(_lib.lib().hsefr_debug_copy(a.data_ptr(), b.data_ptr(), n, _lib.current_stream_ptr())), iters=10)
                    row.append(2.0 * n / med / 1e3)
                print("u=%d ntld=%d ntst=%d wg/cu=%-2d " % (u, nt & 1, nt >> 1, wg) + "".join("%8.0f" % r for r in row))
    _lib.lib().hsefr_debug_set(b"copy_variant", 0)


def bench_dwv():
    print("depthwise cache-policy variants: us median; plain / nt-load / nt-store / both")
    g = torch.Generator(device="cuda").manual_seed(0)
    for hw, c, s in DW:
        x = torch.rand((B, hw, hw, c), device="cuda", generator=g) * 6
        w = torch.randn((3, 3, c), device="cuda", generator=g)
        sc = torch.rand((c,), device="cuda", generator=g) + 0.5
        sh = torch.randn((c,), device="cuda", generator=g)
        res = []
        for v in (0, 1, 2, 3):
            _lib.lib().hsefr_debug_set(b"dw_variant", v)
            res.append(timeit(lambda: ops.dwconv3x3(x, w, sc, sh, s))[0])
        _lib.lib().hsefr_debug_set(b"dw_variant", 0)
        print("%3dx%-3d c=%-4d s%d " % (hw, hw, c, s) + "".join("%9.1f" % r for r in res))


def bench_pwd():
    print("GEMM staging: us median, auto tile: register-staged / LDS-DMA")
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = [0.0, 0.0]
    for hw, k, n in PW:
        m = B * hw * hw
        x = torch.rand((m, k), device="cuda", generator=g) * 6
        w = torch.randn((n, k), device="cuda", generator=g) / k ** 0.5
        sh = torch.randn((n,), device="cuda", generator=g)
        res = []
        for dma in (0, 1, 0, 1):
            _lib.lib().hsefr_debug_set(b"pw_dma", dma)
            res.append(timeit(lambda: ops.pwconv1x1(x, w, sh))[0])
        _lib.lib().hsefr_debug_set(b"pw_dma", 1)
        r0, r1 = min(res[0], res[2]), min(res[1], res[3])
        mult = 5 if (hw, k, n) == (12, 512, 512) else 1
        tot[0] += r0 * mult
        tot[1] += r1 * mult
        fl = 2.0 * m * k * n
        print("%-22s %9.1f %9.1f   TF: %6.1f %6.1f" % ("%dx%dx%d" % (m, k, n), r0, r1, fl / r0 / 1e6, fl / r1 / 1e6))
    print("sum over the 13 pointwise layers: %.1f us vs %.1f us" % tuple(tot))


def bench_pws():
    print("split-f16 GEMM vs fp32-MFMA GEMM, batch %d: us median; error = max|y - y64| / max|y64| on the first 4096 rows" % B)
    print("%-22s %9s | %9s %9s %9s %9s %9s | %8s %8s" % ("M x K x N", "f32 auto", "f16s auto", "128x128", "128x64", "64x64", "-", "err f32", "err f16s"))
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = [0.0, 0.0]
    for hw, k, n in PW:
        m = B * hw * hw
        x = torch.rand((m, k), device="cuda", generator=g) * 6
        x[::7, ::3] = 0
        x[1::5, 1::4] *= 1e-3
        w = torch.randn((n, k), device="cuda", generator=g) / k ** 0.5
        sh = torch.randn((n,), device="cuda", generator=g)
        prep = ops.split_weights_device(w, x.device)
        r32 = timeit(lambda: ops.pwconv1x1(x, w, sh))[0]
        res = []
        for tile in (-1, 0, 1, 2, 3):
            if (tile == 0 and n % 128) or tile == 3:
                res.append(float("nan"))
                continue
            _lib.lib().hsefr_debug_set(b"pws_tile", tile)
            res.append(timeit(lambda: ops.pwconv1x1_f16split(x, None, sh, prepared=prep))[0])
        _lib.lib().hsefr_debug_set(b"pws_tile", -1)
        y64 = torch.clamp(x[:4096].double() @ w.double().T + sh.double(), 0, 6)
        e32 = float((ops.pwconv1x1(x, w, sh)[:4096].double() - y64).abs().max() / y64.abs().max())
        e16 = float((ops.pwconv1x1_f16split(x, None, sh, prepared=prep)[:4096].double() - y64).abs().max() / y64.abs().max())
        mult = 5 if (hw, k, n) == (12, 512, 512) else 1
        tot[0] += r32 * mult
        tot[1] += np.nanmin(res) * mult
        by = 4.0 * (m * k + m * n + k * n)
        print("%-22s %9.1f | %9.1f %9.1f %9.1f %9.1f %9.1f | %8.1e %8.1e  best %5.0f GB/s" % ("%dx%dx%d" % (m, k, n), r32, *res, e32, e16, by / np.nanmin(res) / 1e3))
    print("sum over the 13 pointwise layers: f32 %.1f us, split-f16 (best tile) %.1f us" % tuple(tot))


BLOCKS = [(96, 32, 64, 1), (96, 64, 128, 2), (48, 128, 128, 1), (48, 128, 256, 2), (24, 256, 256, 1), (24, 256, 512, 2),
          (12, 512, 512, 1), (12, 512, 1024, 2), (6, 1024, 1024, 1)]


def bench_blk():
    """One MobileNet block three ways: depthwise kernel + split-f16 GEMM, the fp32 fused kernel (C <= 64), the split-f16 fused kernel."""
    print("block (in hw, c -> cout, stride), batch %d: us median | unfused dw + pw(f16s) | fused fp32 | fused f16s: auto, tw8, tw16, bn64, bn256 | err" % B)
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = [0.0, 0.0]
    for hw, c, n, s in BLOCKS:
        x = torch.rand((B, hw, hw, c), device="cuda", generator=g) * 6
        wd = torch.randn((3, 3, c), device="cuda", generator=g) / 3
        dsc = torch.rand((c,), device="cuda", generator=g) + 0.5
        dsh = torch.randn((c,), device="cuda", generator=g) * 0.3
        w = torch.randn((n, c), device="cuda", generator=g) / c ** 0.5
        sh = torch.randn((n,), device="cuda", generator=g)
        prep = ops.split_weights_device(w, x.device)
        t_dw = timeit(lambda: ops.dwconv3x3(x, wd, dsc, dsh, s))[0]
        mid = ops.dwconv3x3(x, wd, dsc, dsh, s)
        t_pw = timeit(lambda: ops.pwconv1x1_f16split(mid, None, sh, prepared=prep))[0]
        ref = ops.pwconv1x1_f16split(mid, None, sh, prepared=prep)
        t_f32 = float("nan")
        if c <= 64:
            t_f32 = timeit(lambda: ops.dwpw_fused(x, wd, dsc, dsh, w, sh, s))[0]
        res = []
        for key, val in ((b"dwpws_tw", 0), (b"dwpws_tw", 8), (b"dwpws_tw", 16), (b"dwpws_bn", 64), (b"dwpws_bn", 256)):
            _lib.lib().hsefr_debug_set(key, val)
            try:
                res.append(timeit(lambda: ops.dwpw_f16split(x, wd, dsc, dsh, None, sh, s, prepared=prep))[0])
            except Exception:
                res.append(float("nan"))
            _lib.lib().hsefr_debug_set(key, 0)
        y = ops.dwpw_f16split(x, wd, dsc, dsh, None, sh, s, prepared=prep)
        err = float((y - ref).abs().max() / ref.abs().max())
        mult = 5 if (hw, c, n) == (12, 512, 512) else 1
        tot[0] += (t_dw + t_pw) * mult
        tot[1] += min(t_dw + t_pw, np.nanmin(res)) * mult
        by = 4.0 * B * (hw * hw * c + (hw // s) ** 2 * n)
        print("%3d c%-4d->%-4d s%d | %7.1f + %7.1f = %7.1f | %7.1f | " % (hw, c, n, s, t_dw, t_pw, t_dw + t_pw, t_f32) +
              " ".join("%7.1f" % r for r in res) + " | %.1e  best fused %5.0f GB/s" % (err, by / np.nanmin(res) / 1e3))
    print("sum over the 13 blocks: unfused %.1f us, best-of per block %.1f us" % tuple(tot))


def bench_stamps():
    """Diagnostic build only (libhsefr built with -DHSEFR_PWS_STAMPS): where a split-f16 GEMM wave spends its cycles."""
    import ctypes
    g = torch.Generator(device="cuda").manual_seed(0)
    for hw, k, n in ((12, 512, 512), (48, 128, 128), (6, 1024, 1024)):
        m = B * hw * hw
        x = torch.rand((m, k), device="cuda", generator=g) * 6
        w = torch.randn((n, k), device="cuda", generator=g) / k ** 0.5
        sh = torch.randn((n,), device="cuda", generator=g)
        prep = ops.split_weights_device(w, x.device)
        for _ in range(5):
            ops.pwconv1x1_f16split(x, None, sh, prepared=prep)
        torch.cuda.synchronize()
        buf = np.zeros((1024, 8, 8), np.uint64)
        _lib.check(_lib.lib().hsefr_debug_read_stamps(0, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
        used = buf[:, :, 7] > 0
        b = buf[used].astype(np.float64)
        names = ["gload issue", "ds_read+mfma issue", "wait+convert+ds_write", "barrier", "epilogue/loop", "step top"]
        tot = b[:, 6]
        print("%dx%dx%d: %d waves, lifetime cycles mean %.0f min %.0f max %.0f, steps/wave mean %.1f" %
              (m, k, n, len(b), tot.mean(), tot.min(), tot.max(), b[:, 7].mean()))
        for i, nm in enumerate(names):
            print("   %-24s %6.1f %% of lifetime, %7.0f cycles per step" % (nm, 100 * (b[:, i] / tot).mean(), (b[:, i] / b[:, 7]).mean()))


def bench_stem():
    g = torch.Generator(device="cuda").manual_seed(0)
    for hw in (192, 224):
        x = (torch.rand((B, hw, hw, 3), device="cuda", generator=g) - 0.5) * 256
        cw = torch.randn((3, 3, 3, 32), device="cuda", generator=g) * 0.02
        csh = torch.randn((32,), device="cuda", generator=g)
        wd = torch.randn((3, 3, 32), device="cuda", generator=g) / 3
        dsc = torch.rand((32,), device="cuda", generator=g) + 0.5
        dsh = torch.randn((32,), device="cuda", generator=g) * 0.3
        w = torch.randn((64, 32), device="cuda", generator=g) / 32 ** 0.5
        sh = torch.randn((64,), device="cuda", generator=g)
        t_c = timeit(lambda: ops.conv3x3_c3(x, cw, csh, 2))[0]
        mid = ops.conv3x3_c3(x, cw, csh, 2)
        t_b = timeit(lambda: ops.dwpw_fused(mid, wd, dsc, dsh, w, sh, 1))[0]
        prep = ops.split_weights_device(w, x.device)
        t_s = timeit(lambda: ops.stem_fused(x, cw, csh, wd, dsc, dsh, None, sh, prepared=prep))[0]
        by = 4.0 * B * (hw * hw * 3 + (hw // 2) ** 2 * 64)
        print("stem %d: conv1 %.1f + fused block %.1f = %.1f us | fused stem %.1f us  (%.0f GB/s of in+out)" %
              (hw, t_c, t_b, t_c + t_b, t_s, by / t_s / 1e3))
        wd2 = torch.randn((3, 3, 64), device="cuda", generator=g) / 3
        d2sc = torch.rand((64,), device="cuda", generator=g) + 0.5
        d2sh = torch.randn((64,), device="cuda", generator=g) * 0.3
        w2 = torch.randn((128, 64), device="cuda", generator=g) / 8
        sh2 = torch.randn((128,), device="cuda", generator=g)
        prep2 = ops.split_weights_device(w2, x.device)
        y1 = ops.stem_fused(x, cw, csh, wd, dsc, dsh, None, sh, prepared=prep)
        t_b2 = timeit(lambda: ops.dwpw_fused(y1, wd2, d2sc, d2sh, w2, sh2, 2))[0]
        t_s2 = timeit(lambda: ops.stem2_fused(x, cw, csh, wd, dsc, dsh, None, sh, wd2, d2sc, d2sh, prepared=prep))[0]
        y2 = ops.stem2_fused(x, cw, csh, wd, dsc, dsh, None, sh, wd2, d2sc, d2sh, prepared=prep)
        t_p2 = timeit(lambda: ops.pwconv1x1_f16split(y2, None, sh2, prepared=prep2))[0]
        print("   + block 2: stem %.1f + fused block 2 %.1f = %.1f us | stem+dw2 %.1f + pointwise 2 %.1f = %.1f us" %
              (t_s, t_b2, t_s + t_b2, t_s2, t_p2, t_s2 + t_p2))


def bench_stemstamps():
    """Diagnostic build only (-DHSEFR_STEM_STAMPS): where a fused-stem wave spends its cycles."""
    import ctypes
    g = torch.Generator(device="cuda").manual_seed(0)
    hw = 192
    x = (torch.rand((B, hw, hw, 3), device="cuda", generator=g) - 0.5) * 256
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
    which = os.environ.get("KB_STEM", "2")
    if "KB_STEM4_GRID" in os.environ:
        _lib.check(_lib.lib().hsefr_debug_set(b"stem4_grid", int(os.environ["KB_STEM4_GRID"])))
    for _ in range(4):
        if which == "1":
            ops.stem_fused(x, cw, csh, wd, dsc, dsh, w, sh)
        elif which == "3":
            ops.stem3_fused(x * 0.99, cw, csh, wd, dsc, dsh, w, sh, wd2, d2sc, d2sh)
        elif which == "4":
            ops.stem4_fused(x * 0.99, cw, csh, wd, dsc, dsh, w, sh, wd2, d2sc, d2sh)
        elif which == "4u8":
            ops.stem4_fused((x * 0.99 + 128).clamp(0, 255).to(torch.uint8), cw, csh, wd, dsc, dsh, w, sh, wd2, d2sc, d2sh,
                            u8_mean_bgr=(103.939, 116.779, 123.68))
        else:
            ops.stem2_fused(x, cw, csh, wd, dsc, dsh, w, sh, wd2, d2sc, d2sh)
    torch.cuda.synchronize()
    buf = np.zeros((512 * 4, 10), np.uint64)
    _lib.check(_lib.lib().hsefr_debug_read_stamps(1, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
    b = buf[buf[:, 9] > 0].astype(np.float64)
    names = ["cursor + gather issue", "B conv1 mfma + region write", "barriers", "C depthwise 1 from LDS", "D pointwise mfma (+ patch write)",
             "E epilogue / depthwise 2 + stores", "scatter (wait gather)", "-"]
    if which == "3":
        names = ["cursor", "A' im2col from the window", "barriers (+ park next window)", "B conv1 mfma + region write", "C depthwise 1", "D pointwise mfma (+ patch write)",
                 "E depthwise 2 + stores", "window loads issue"]
    if which in ("4", "4u8"):
        names = ["cursor", "park next window + border tables", "barriers", "B conv1 from the window + region write", "C depthwise 1 (+ next window loads issue)",
                 "D pointwise mfma (+ patch write)", "E depthwise 2 + stores", "-"]
    print("%d waves, lifetime mean %.0f cycles, patches/wave %.1f -> %.0f cycles per patch" % (len(b), b[:, 8].mean(), b[:, 9].mean(), (b[:, 8] / b[:, 9]).mean()))
    for i, nm in enumerate(names):
        print("   %-26s %5.1f %%  %7.0f cycles per patch" % (nm, 100 * (b[:, i] / b[:, 8]).mean(), (b[:, i] / b[:, 9]).mean()))


def bench_blkstamps():
    """Diagnostic build only (-DHSEFR_STEM_STAMPS): where a wave of the v2 fused block spends its cycles."""
    import ctypes
    g = torch.Generator(device="cuda").manual_seed(0)
    for hw, c, n in ((48, 128, 128), (24, 256, 256)):
        x = torch.rand((B, hw, hw, c), device="cuda", generator=g) * 6
        wd = torch.randn((3, 3, c), device="cuda", generator=g) / 3
        dsc = torch.rand((c,), device="cuda", generator=g) + 0.5
        dsh = torch.randn((c,), device="cuda", generator=g) * 0.3
        w = torch.randn((n, c), device="cuda", generator=g) / c ** 0.5
        sh = torch.randn((n,), device="cuda", generator=g)
        prep = ops.split_weights_device(w, x.device)
        for _ in range(4):
            ops.dwpw_f16split(x, wd, dsc, dsh, None, sh, 1, prepared=prep)
        torch.cuda.synchronize()
        buf = np.zeros((512 * 4, 10), np.uint64)
        _lib.check(_lib.lib().hsefr_debug_read_stamps(1, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
        names = ["wait vmcnt", "DMA issue", "depthwise (LDS -> A tile)", "barrier", "MFMA", "epilogue"]
        for role, rows in (("producers (v3) / all waves (v2)", buf[:1024]), ("consumers (v3)", buf[1024:])):
            b = rows[rows[:, 9] > 0].astype(np.float64)
            if not len(b):
                continue
            print("%d c%d->%d %s: %d waves, lifetime mean %.0f cycles, patches/wave %.1f -> %.0f cycles per patch" %
                  (hw, c, n, role, len(b), b[:, 8].mean(), b[:, 9].mean(), (b[:, 8] / b[:, 9]).mean()))
            for i, nm in enumerate(names):
                print("   %-26s %5.1f %%  %7.0f cycles per patch" % (nm, 100 * (b[:, i] / b[:, 8]).mean(), (b[:, i] / b[:, 9]).mean()))


def bench_pwa():
    print("GEMM ablations (timing only), 128x64 tile: us median: real / no-global-loads / no-stores / neither")
    g = torch.Generator(device="cuda").manual_seed(0)
    for hw, k, n in PW[2:]:
        m = B * hw * hw
        x = torch.rand((m, k), device="cuda", generator=g) * 6
        w = torch.randn((n, k), device="cuda", generator=g) / k ** 0.5
        sh = torch.randn((n,), device="cuda", generator=g)
        _lib.lib().hsefr_debug_set(b"pw_tile", 1)
        res = []
        for ab in (0, 1, 2, 3):
            _lib.lib().hsefr_debug_set(b"pw_ablate", ab)
            res.append(timeit(lambda: ops.pwconv1x1(x, w, sh))[0])
        _lib.lib().hsefr_debug_set(b"pw_ablate", 0)
        _lib.lib().hsefr_debug_set(b"pw_tile", -1)
        fl = 2.0 * m * k * n
        print("%-22s" % ("%dx%dx%d" % (m, k, n)) + "".join("%9.1f" % r for r in res) + "   TF: " + " ".join("%6.1f" % (fl / r / 1e6) for r in res))


def bench_clock():
    """Shader clock and fp32-MFMA rate the chip sustains with every SIMD issuing MFMAs back to back."""
    for mode, blocks, label in ((0, 256, "fp32 1 wave/SIMD"), (0, 512, "fp32 2 waves/SIMD"), (0, 768, "fp32 3 waves/SIMD"),
                                (1, 256, "f16 1 wave/SIMD"), (1, 512, "f16 2 waves/SIMD"), (1, 768, "f16 3 waves/SIMD"),
                                (2, 256, "f16 16x16x32 1 w/SIMD"), (2, 512, "f16 16x16x32 2 w/SIMD")):
        _lib.lib().hsefr_debug_set(b"clock_mode", mode)
        iters = 200000 if mode == 0 else 400000
        out = torch.zeros((blocks * 3,), dtype=torch.int64, device="cuda")
        for _ in range(2):
            _lib.check(_lib.lib().hsefr_debug_clock_probe(out.data_ptr(), blocks, iters, _lib.current_stream_ptr()))
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        _lib.check(_lib.lib().hsefr_debug_clock_probe(out.data_ptr(), blocks, iters, _lib.current_stream_ptr()))
        ev1.record()
        torch.cuda.synchronize()
        o = out.cpu().numpy().reshape(blocks, 3)
        clk = np.median(o[:, 0] / o[:, 1]) * 100e6
        ms = ev0.elapsed_time(ev1)
        tf = blocks * 4 * iters * (4 * 4096.0 if mode == 0 else (4 * 32768.0 if mode == 1 else 8 * 16384.0)) / (ms * 1e-3) / 1e12
        print("clock probe %-18s: shader clock %.3f GHz (min %.3f max %.3f), %.2f ms, %.1f TFLOP/s MFMA" %
              (label, clk / 1e9, (o[:, 0] / o[:, 1]).min() / 10, (o[:, 0] / o[:, 1]).max() / 10, ms, tf))
    _lib.lib().hsefr_debug_set(b"clock_mode", 0)


if __name__ == "__main__":
    what = sys.argv[1:] or ["pw", "dw", "c3"]
    for w in what:
        {"pw": bench_pw, "dw": bench_dw, "c3": bench_c3, "copy": bench_copy, "dwv": bench_dwv, "clock": bench_clock, "pwa": bench_pwa, "pwd": bench_pwd, "pws": bench_pws, "blk": bench_blk, "stamps": bench_stamps, "stem": bench_stem, "stemstamps": bench_stemstamps, "blkstamps": bench_blkstamps}[w]()
