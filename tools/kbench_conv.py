#!/usr/bin/env python3
"""bf16 convolutions of ResNet-50 @ batch 128 one at a time: the LDS-DMA implicit GEMM (csrc/conv_dma_bf16.hip) per tile height
RB against the register-staged kernels (development library: knobs cd_rb / cd_off).
    HSEFR_LIB=hse_facerec_tf_amd/libhsefr_dev.so python tools/kbench_conv.py [layer ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hse_facerec_tf_amd import _lib, ops

B = int(os.environ.get("KB_BATCH", "128"))
ITERS = int(os.environ.get("KB_ITERS", "20"))
# name: (hw, cin, cout, k, stride, residual)
LAYERS = {
    "c2_3x3": (56, 64, 64, 3, 1, False), "c2_red": (56, 256, 64, 1, 1, False), "c2_inc": (56, 64, 256, 1, 1, True),
    "c3_3x3": (28, 128, 128, 3, 1, False), "c3_red": (28, 512, 128, 1, 1, False), "c3_inc": (28, 128, 512, 1, 1, True),
    "c4_3x3": (14, 256, 256, 3, 1, False), "c4_red": (14, 1024, 256, 1, 1, False), "c4_inc": (14, 256, 1024, 1, 1, True),
    "c5_3x3": (7, 512, 512, 3, 1, False), "c5_red": (7, 2048, 512, 1, 1, False), "c5_inc": (7, 512, 2048, 1, 1, True),
    "c4_proj": (28, 512, 1024, 1, 2, False), "c3_proj": (56, 256, 512, 1, 2, False), "c5_proj": (14, 1024, 2048, 1, 2, False),
    "c3_red2": (56, 256, 128, 1, 2, False), "c4_red2": (28, 512, 256, 1, 2, False), "c5_red2": (14, 1024, 512, 1, 2, False),
    "c2_proj": (56, 64, 256, 1, 1, False),
}


def timeit(fn, iters=ITERS, warm=3):
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
    return ts[len(ts) // 2] * 1e3


def knob(key, v):
    _lib.check(_lib.lib().hsefr_debug_set(key.encode(), int(v)), "hsefr_debug_set")


def main():
    names = sys.argv[1:] or list(LAYERS)
    g = torch.Generator(device="cuda").manual_seed(0)
    for name in names:
        hw, c, cout, k, s, res = LAYERS[name]
        x = (torch.rand((B, hw, hw, c), device="cuda", generator=g) * 2).to(torch.bfloat16)
        w = (torch.randn((cout, k * k * c), device="cuda", generator=g) / (k * k * c) ** 0.5).to(torch.bfloat16)
        sc = torch.ones(cout, device="cuda")
        sh = torch.zeros(cout, device="cuda")
        oh = (hw + 2 * (k // 2) - k) // s + 1
        r = (torch.rand((B, oh, oh, cout), device="cuda", generator=g)).to(torch.bfloat16) if res else None
        fn = lambda: ops.conv_bf16(x, w, sc, sh, k, k, stride=s, pad=k // 2, res=r)
        flops = 2.0 * B * oh * oh * cout * k * k * c
        knob("cd_off", 1)
        knob("w3_off", 1)
        knob("w2_off", 1)
        t_old = timeit(fn)
        knob("cd_off", 2)
        out = "%-8s old %6.1f us (%4.0f TF) | dma" % (name, t_old, flops / t_old / 1e6)
        rbs = range(4, 10) if cout % 128 == 0 else range(2, 6)
        for rb in rbs:
            knob("cd_rb", rb)
            t = timeit(fn)
            out += "  RB%d %6.1f" % (rb, t)
        knob("cd_rb", 0)
        t = timeit(fn)
        if k == 3 and s == 1:
            knob("w3_off", 2)
            out += "  | window %6.1f" % timeit(fn)
        if k == 3 and s == 1:
            knob("w3_off", 1)
            knob("w2_off", 2)
            try:
                out += "  | w2 %6.1f" % timeit(fn)
            except Exception as e:          # (shapes the four-wave window kernel does not cover: cout % 128 on the narrow maps)
                out += "  | w2   n/a"
        if k == 1 and s == 1:
            knob("w4_off", 1)
            knob("cd_off", 1)
            for tile in (1, 2):
                knob("c11_tile", tile)
                out += "  | c11 %s %6.1f" % ("128x64" if tile == 1 else "128x128", timeit(fn))
            knob("c11_tile", 0)
            knob("c11_bres", 0)
            out += "  | c11 weights reloaded %6.1f" % timeit(fn)
            knob("c11_bres", 1)
            knob("cd_off", 2)
        if k == 1:
            knob("w4_off", 2)
            out += "  | w4 %6.1f" % timeit(fn)
            knob("w4_off", 0)
        knob("cd_off", 0)
        knob("w3_off", 0)
        knob("w2_off", 0)
        t_prod = timeit(fn)
        out += "  | auto %6.1f us (%4.0f TF) | product dispatch %6.1f us" % (t, flops / t / 1e6, t_prod)
        print(out, flush=True)


if __name__ == "__main__":
    main()
