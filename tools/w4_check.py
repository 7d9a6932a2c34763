#!/usr/bin/env python3
"""conv1x1_w4_bf16 (csrc/conv1x1_w4_bf16.hip) against the kernels it replaces, on the development library:
    HSEFR_LIB=libhsefr_dev.so python tools/w4_check.py
every shape once forced through the new kernel (w4_off = 2) and once without it: small-integer inputs must agree exactly, random ones
within a bf16 ulp of the addends (other K order)."""
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
    #          n, h, w, c, cout, stride, res
    shapes = [(2, 14, 14, 64, 128, 1, False), (3, 14, 14, 256, 1024, 1, True), (1, 15, 16, 128, 64, 1, False), (2, 13, 12, 64, 192, 1, True),
              (2, 28, 28, 512, 256, 2, False), (1, 9, 25, 64, 128, 2, False), (1, 56, 56, 256, 64, 1, False), (5, 7, 7, 2048, 512, 1, False),
              (5, 7, 7, 512, 2048, 1, True), (128, 14, 14, 256, 1024, 1, True), (128, 14, 14, 1024, 256, 1, False), (128, 56, 56, 64, 256, 1, True),
              (128, 56, 56, 256, 64, 1, False), (128, 56, 56, 256, 128, 2, False), (128, 7, 7, 512, 2048, 1, True), (128, 28, 28, 128, 512, 1, True)]
    for (n, h, w, c, cout, st, res) in shapes:
        oh, ow = (h - 1) // st + 1, (w - 1) // st + 1
        for ints in (True, False):
            if ints:
                x = torch.randint(-2, 3, (n, h, w, c), device="cuda", generator=g).to(torch.bfloat16)
                wt = torch.randint(-1, 2, (cout, c), device="cuda", generator=g).to(torch.bfloat16)
                sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
                r = torch.randint(-3, 4, (n, oh, ow, cout), device="cuda", generator=g).to(torch.bfloat16) if res else None
                act = 0
            else:
                x = (torch.rand((n, h, w, c), device="cuda", generator=g) * 2).to(torch.bfloat16)
                wt = (torch.randn((cout, c), device="cuda", generator=g) / c ** 0.5).to(torch.bfloat16)
                sc = torch.rand(cout, device="cuda", generator=g) + 0.5
                sh = torch.randn(cout, device="cuda", generator=g) * 0.1
                r = torch.rand((n, oh, ow, cout), device="cuda", generator=g).to(torch.bfloat16) if res else None
                act = 1
            knob("w4_off", 1)
            ref = ops.conv_bf16(x, wt, sc, sh, 1, 1, st, 0, r, act).float()
            knob("w4_off", 2)
            got = [ops.conv_bf16(x, wt, sc, sh, 1, 1, st, 0, r, act).float() for _ in range(3)]
            knob("w4_off", 0)
            same_runs = all(torch.equal(got[0], o) for o in got[1:])
            d = (got[0] - ref).abs()
            if ints:
                ok = bool(torch.equal(got[0], ref))
            else:
                tol = 2.0 ** -7 * ref.abs() + 2.0 ** -8 * float(ref.abs().max())
                ok = bool((d <= tol).all())
            print("%-36s %-5s %s run-to-run %s  max|d| %.4g  differing %d of %d" %
                  ((n, h, w, c, cout, st, res), "ints" if ints else "rand", "OK " if ok else "BAD", same_runs, float(d.max()), int((d > 0).sum()), d.numel()), flush=True)
            bad += (not ok) + (not same_runs)
            if not ok:
                idx = torch.nonzero(d > (0 if ints else tol))[:8]
                for i in idx.tolist():
                    print("    at", i, "got", float(got[0][tuple(i)]), "want", float(ref[tuple(i)]))
    print("FAILED" if bad else "ALL OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
