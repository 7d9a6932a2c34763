#!/usr/bin/env python3
"""conv3x3_w2_bf16 (csrc/conv3x3_w2_bf16.hip) against the kernels it replaces, on the development library:
    HSEFR_LIB=libhsefr_dev.so python tools/w2_check.py
every shape once forced through the new kernel (w2_off = 2) and once with it disabled (w2_off = 1): the two may differ by one
bf16 ulp where a sum sits on a rounding boundary (other K order); small-integer inputs must agree exactly."""
import os
import sys

os.environ.setdefault("HSEFR_LIB", "libhsefr_dev.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hse_facerec_tf_amd import _lib, ops


def knob(key, v):
    _lib.check(_lib.lib().hsefr_debug_set(key.encode(), int(v)), "hsefr_debug_set")


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    bad = 0
    shapes = [(2, 14, 14, 64, 128, False), (3, 14, 14, 256, 256, True), (1, 15, 16, 128, 128, False), (2, 13, 12, 64, 128, True),
              (2, 28, 28, 128, 128, False), (1, 9, 25, 64, 128, True), (1, 30, 32, 64, 256, False),
              (1, 56, 56, 64, 64, False), (2, 6, 50, 128, 64, True), (1, 7, 64, 64, 192, False),
              (3, 7, 7, 512, 512, False), (5, 6, 7, 64, 64, True), (2, 7, 6, 128, 192, False), (130, 7, 7, 128, 512, True), (128, 7, 7, 512, 512, False),
              (260, 14, 14, 64, 128, False), (70, 28, 28, 64, 128, True), (20, 56, 56, 64, 64, False),
              (128, 14, 14, 256, 256, False), (128, 28, 28, 128, 128, False), (128, 56, 56, 64, 64, False)]
    for (n, h, w, c, cout, res) in shapes:
        for ints in (True, False):
            if ints:
                x = torch.randint(-2, 3, (n, h, w, c), device="cuda", generator=g).to(torch.bfloat16)
                wt = torch.randint(-1, 2, (cout, 9 * c), device="cuda", generator=g).to(torch.bfloat16)
                sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
                r = torch.randint(-3, 4, (n, h, w, cout), device="cuda", generator=g).to(torch.bfloat16) if res else None
                act = 0
            else:
                x = (torch.rand((n, h, w, c), device="cuda", generator=g) * 2).to(torch.bfloat16)
                wt = (torch.randn((cout, 9 * c), device="cuda", generator=g) / (9 * c) ** 0.5).to(torch.bfloat16)
                sc = torch.rand(cout, device="cuda", generator=g) + 0.5
                sh = torch.randn(cout, device="cuda", generator=g) * 0.1
                r = torch.rand((n, h, w, cout), device="cuda", generator=g).to(torch.bfloat16) if res else None
                act = 1
            knob("w2_off", 1)
            ref = ops.conv_bf16(x, wt, sc, sh, 3, 3, 1, 1, r, act).float()
            knob("w2_off", 2)
            got = [ops.conv_bf16(x, wt, sc, sh, 3, 3, 1, 1, r, act).float() for _ in range(3)]
            knob("w2_off", 0)
            same_runs = all(torch.equal(got[0], o) for o in got[1:])
            d = (got[0] - ref).abs()
            if ints:
                ok = bool(torch.equal(got[0], ref))
            else:
                tol = 2.0 ** -7 * ref.abs() + 2.0 ** -9 * float(ref.abs().max())
                ok = bool((d <= tol).all())
            nbad = int((d > 0).sum())
            print("%-28s %-5s %s run-to-run %s  max|d| %.4g  differing %d of %d" %
                  ((n, h, w, c, cout, res), "ints" if ints else "rand", "OK " if ok else "BAD", same_runs, float(d.max()), nbad, d.numel()), flush=True)
            bad += (not ok) + (not same_runs)
            if not ok:
                idx = torch.nonzero(d > (0 if ints else tol))[:8]
                for i in idx.tolist():
                    print("    at", i, "got", float(got[0][tuple(i)]), "want", float(ref[tuple(i)]))
    print("FAILED" if bad else "ALL OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
