"""ResNet-50 (VGGFace2 `resnet50_ft`) on the bf16 MFMA path.

The reference reaches this network through ``TensorFlowInference('models/vgg2_resnet.pb',
input_tensor='input:0', output_tensor='pool5_7x7_s1:0', convert2BGR=True, imageNetUtilsMean=False)``
(facerec_test.py:213); the 95 MB file itself is one of the reference's missing blobs
(.MISSING_LARGE_BLOBS), so the topology is built here from the Caffe `resnet50_ft` layer names that
the output tensor name (`pool5/7x7_s1`) belongs to:

    conv1/7x7_s2 (64, /2, pad 3) + BN + ReLU -> pool1/3x3_s2 (max, Caffe ceil mode: 112 -> 56)
    -> conv2_1..3 | conv3_1..4 | conv4_1..6 | conv5_1..3   bottlenecks (mid, out) = (64,256) (128,512)
       (256,1024) (512,2048); each = 1x1_reduce(+BN+ReLU) -> 3x3(+BN+ReLU) -> 1x1_increase(+BN),
       + shortcut (identity, or 1x1_proj+BN on the first block of a stage), ReLU;
       the stride-2 of stages 3-5 sits on the FIRST 1x1 (`_reduce`) and on `_proj`
    -> pool5/7x7_s1 (average) -> 2048-D

Weights are a dict {name: array}: ``<conv>/kernel`` [kh,kw,cin,cout] fp32 (TF HWIO) and the folded
BatchNorm ``<conv>/scale``, ``<conv>/shift`` [cout] (scale = gamma/sqrt(var+eps), shift = beta -
mean*scale).  ``synthetic_weights`` produces a seeded random set of the right shapes for benchmarks
(there is no real file to load); a user who has vgg2_resnet.pb's tensors can pass them in this form.
Activations are bf16 in HBM, accumulation is fp32 (BASELINE config 3).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

from . import lowering
from .lowering import (ACT_NONE, ACT_RELU, OP_CONV_BF16, OP_CONV_F32, OP_GAP, OP_GAP_BF16, OP_MAXPOOL_F32, OP_MAXPOOL_BF16, OP_STEM7X7_BF16, OUT_FEATURES,
                       Layer, Plan, assign_buffers)

STAGES = [("conv2", 3, 64, 256, 1), ("conv3", 4, 128, 512, 2), ("conv4", 6, 256, 1024, 2), ("conv5", 3, 512, 2048, 2)]


def conv_names() -> List[Tuple[str, int, int, int, int]]:
    """[(name, k, cin, cout, stride)] of every convolution, in execution order."""
    out = [("conv1_7x7_s2", 7, 3, 64, 2)]
    cin = 64
    for stage, blocks, mid, cout, stride in STAGES:
        for b in range(1, blocks + 1):
            s = stride if b == 1 else 1
            pre = "%s_%d" % (stage, b)
            out.append((pre + "_1x1_reduce", 1, cin, mid, s))
            out.append((pre + "_3x3", 3, mid, mid, 1))
            out.append((pre + "_1x1_increase", 1, mid, cout, 1))
            if b == 1:
                out.append((pre + "_1x1_proj", 1, cin, cout, s))
            cin = cout
    return out


def synthetic_weights(seed: int = 123) -> Dict[str, np.ndarray]:
    """He-normal kernels; folded-BN scale ~ 1 and shift ~ 0 with a little spread; the last BN of every
    bottleneck is damped so the residual sum keeps O(1) magnitude through 16 blocks."""
    rs = np.random.RandomState(seed)
    w: Dict[str, np.ndarray] = {}
    for name, k, cin, cout, _ in conv_names():
        fan_in = k * k * cin
        w[name + "/kernel"] = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        damp = 0.3 if name.endswith("_increase") else 1.0
        w[name + "/scale"] = (damp * rs.uniform(0.8, 1.2, cout)).astype(np.float32)
        w[name + "/shift"] = (0.05 * rs.randn(cout)).astype(np.float32)
    return w


def to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns (uint16), round-to-nearest-even."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def pack_conv_weight(kernel_hwio: np.ndarray) -> np.ndarray:
    """[kh,kw,cin,cout] -> [cout][kh*kw*cin] bf16 bits (k = (kh*KW + kw)*cin + ci)."""
    kh, kw, cin, cout = kernel_hwio.shape
    return to_bf16_bits(kernel_hwio.reshape(kh * kw * cin, cout).T)


def pack_stem_weight(kernel_hwio: np.ndarray) -> np.ndarray:
    """[7,7,3,64] -> [64][256] bf16 bits with k = dy*32 + dx*3 + ci, zero padded."""
    assert kernel_hwio.shape == (7, 7, 3, 64)
    img = np.zeros((64, 8, 32), np.float32)
    img[:, :7, :21] = kernel_hwio.reshape(7, 21, 64).transpose(2, 0, 1)
    return to_bf16_bits(img.reshape(64, 256))


def build_plan(weights: Dict[str, np.ndarray], input_hw: Tuple[int, int] = (224, 224), pool: str = "caffe", dtype: str = "bf16",
               fuse: bool = True, pair: bool = True, subsample: bool = True) -> Plan:
    """pool='caffe': pad-0 ceil-mode max-pool (112 -> 56); pool='valid': keras_vggface's valid pool (112 -> 55).
    dtype='bf16': the bf16-MFMA kernels (BASELINE config 3's throughput mode); dtype='f32': the same layers on the exact-fp32
    general kernels (OP_CONV_F32 / OP_MAXPOOL_F32 / OP_GAP) -- the fp32-grade mode, 1e-4 against the fp64 oracle.
    fuse (bf16 only): conv1 + pool1 run as one kernel (lowering.fuse_stem_pool) and the four projected shortcuts run inside their
    block's increase layer (lowering.fuse_proj); False keeps every layer's tensor.
    pair (with fuse): the 56-pixel stage's increase layers run in one launch with the next block's reduce layer (lowering.mark_pairs).
    subsample (with fuse): the last block of the 56-, 28- and 14-pixel stages computes its 3x3 and increase layers only at the pixels the
    next stage's stride-2 layers read (lowering.subsample_stage_tails): same features, three layers' tensors become [::2, ::2] of the graph's."""
    if dtype not in ("bf16", "f32"):
        raise ValueError("dtype must be 'bf16' or 'f32', not %r" % (dtype,))
    f32 = dtype == "f32"
    H, W = input_hw
    layers: List[Layer] = []

    def add(L: Layer) -> int:
        layers.append(L)
        return len(layers) - 1

    def conv_out(h, k, s, pad):
        return (h + 2 * pad - k) // s + 1

    oh, ow = conv_out(H, 7, 2, 3), conv_out(W, 7, 2, 3)
    cur = add(Layer(OP_CONV_F32 if f32 else OP_STEM7X7_BF16, "conv1_7x7_s2", -1, (H, W, 3), (oh, ow, 64),
                    w=weights["conv1_7x7_s2/kernel"].astype(np.float32) if f32 else pack_stem_weight(weights["conv1_7x7_s2/kernel"]),
                    scale=weights["conv1_7x7_s2/scale"], shift=weights["conv1_7x7_s2/shift"], act=ACT_RELU, kh=7, kw=7,
                    stride=2, pad_t=3, pad_l=3))
    if pool == "caffe":
        ph, pw = -(-(oh - 3) // 2) + 1, -(-(ow - 3) // 2) + 1
    elif pool == "valid":
        ph, pw = (oh - 3) // 2 + 1, (ow - 3) // 2 + 1
    else:
        raise ValueError(pool)
    cur = add(Layer(OP_MAXPOOL_F32 if f32 else OP_MAXPOOL_BF16, "pool1_3x3_s2", cur, (oh, ow, 64), (ph, pw, 64), kh=3, kw=3, stride=2))
    h, w_, cin = ph, pw, 64

    def conv(name, src, hwc, k, cout, stride, act, res=-1):
        hh, ww, cc = hwc
        pad = (k - 1) // 2
        o = (conv_out(hh, k, stride, pad), conv_out(ww, k, stride, pad), cout)
        return add(Layer(OP_CONV_F32 if f32 else OP_CONV_BF16, name, src, hwc, o,
                         w=weights[name + "/kernel"].astype(np.float32) if f32 else pack_conv_weight(weights[name + "/kernel"]),
                         scale=weights[name + "/scale"], shift=weights[name + "/shift"], act=act, kh=k, kw=k,
                         stride=stride, pad_t=pad, pad_l=pad, res=res))

    for stage, blocks, mid, cout, stride in STAGES:
        for b in range(1, blocks + 1):
            s = stride if b == 1 else 1
            pre = "%s_%d" % (stage, b)
            x_in, x_shape = cur, (h, w_, cin)
            r = conv(pre + "_1x1_reduce", x_in, x_shape, 1, mid, s, ACT_RELU)
            t = conv(pre + "_3x3", r, layers[r].out_shape, 3, mid, 1, ACT_RELU)
            sc = conv(pre + "_1x1_proj", x_in, x_shape, 1, cout, s, ACT_NONE) if b == 1 else x_in
            cur = conv(pre + "_1x1_increase", t, layers[t].out_shape, 1, cout, 1, ACT_RELU, res=sc)
            h, w_, cin = layers[cur].out_shape
    gap = add(Layer(OP_GAP if f32 else OP_GAP_BF16, "pool5_7x7_s1", cur, (h, w_, cin), (1, 1, cin)))
    for L in layers:
        L.sealed = True
    if fuse and f32 and subsample:
        lowering.subsample_stage_tails(layers, [gap])
    if fuse and not f32:
        layers, remap = lowering.fuse_stem_pool(layers, [gap])
        gap = remap[gap]
        layers, remap = lowering.fuse_proj(layers, [gap])
        gap = remap[gap]
        if subsample:
            lowering.subsample_stage_tails(layers, [gap])
        if pair:
            lowering.mark_pairs(layers)
            if subsample:
                lowering.compact_pair_outputs(layers, [gap])
    buffers = assign_buffers(layers, {gap})
    names = {L.name: i for i, L in enumerate(layers)}
    return Plan(layers, (H, W, 3), buffers, {OUT_FEATURES: (gap, cin)}, names)


def _graph_scale(L) -> float:
    """Layers of a stage's last block compute every s-th pixel of their graph tensor (lowering.subsample_stage_tails): the ALGORITHMIC figures
    below stay the graph's -- the work the reference does -- whatever the plan skips."""
    if L.graph_hw is None:
        return 1.0
    return (L.graph_hw[0] * L.graph_hw[1]) / float(L.out_shape[0] * L.out_shape[1])


def flops_per_image(plan: Plan) -> int:
    tot = 0
    for L in plan.layers:
        if L.kind == lowering.OP_STEM7X7_POOL_BF16:
            tot += Plan.layer_flops(L)
        if L.kind in (OP_CONV_BF16, OP_STEM7X7_BF16, OP_CONV_F32):
            # (a projected shortcut folded into its increase layer counts there; an OUT_SUB2 layer computes -- and counts -- every pixel already)
            tot += int(round(Plan.layer_flops(L) * (1.0 if L.flags & lowering.OPF_OUT_SUB2 else _graph_scale(L))))
    return tot


def executed_flops_per_image(plan: Plan) -> int:
    """What the plan's kernels compute (flops_per_image minus the pixels subsample_stage_tails skips)."""
    return sum(Plan.layer_flops(L) for L in plan.layers if L.kind in (lowering.OP_STEM7X7_POOL_BF16, OP_CONV_BF16, OP_STEM7X7_BF16, OP_CONV_F32))


def activation_bytes_per_image(plan: Plan) -> int:
    """The layer-wise minimum traffic of the GRAPH: every layer reads its input once and writes its output once, whatever the plan
    fuses or skips."""
    tot = 0
    for L in plan.layers:
        in_b = 4 if L.kind in (OP_STEM7X7_BF16, lowering.OP_STEM7X7_POOL_BF16) else 2
        src = plan.layers[L.src] if L.src >= 0 else None
        in_elems = int(np.prod(L.in_shape))
        if src is not None and src.graph_hw is not None:      # in the graph this layer read the full-size tensor
            in_elems = src.graph_hw[0] * src.graph_hw[1] * src.out_shape[2]
        tot += in_elems * in_b + int(round(L.out_bytes * _graph_scale(L)))
        if L.res >= 0 and L.proj is None:
            tot += int(round(L.out_bytes * _graph_scale(L)))      # (the residual has the layer's own shape in the graph)
        elif L.res >= 0:
            # the UNFUSED pair's traffic (the layer-wise figure SURVEY 8d prices the network with, unchanged by the fusion): the
            # projection reads the block input and writes its tensor, the increase layer reads it back
            c2, s2, h2, w2 = L.proj
            R = plan.layers[L.res]
            if R.graph_hw is not None:
                h2, w2 = R.graph_hw
            oh, ow, cout = L.out_shape
            tot += h2 * w2 * c2 * 2 + 2 * oh * ow * cout * 2
    return tot


class ResNet50Extractor:
    """Batched extractor with the TensorFlowInference surface used by the hot loop (w, h, extract_batch,
    close_session); preprocessing flags are those of facerec_test.py:213 (BGR, VGGFace2 mean)."""

    def __init__(self, weights: Optional[Dict[str, np.ndarray]] = None, input_size: Tuple[int, int] = (224, 224),
                 max_batch: int = 128, device: Optional[int] = None, pool: str = "caffe", seed: int = 123, dtype: str = "bf16",
                 fuse: bool = True):
        from .engine import Engine
        self.w, self.h = input_size
        self.convert2BGR, self.imageNetUtilsMean = True, False
        self.plan = build_plan(weights if weights is not None else synthetic_weights(seed), (self.h, self.w), pool, dtype, fuse)
        self.dtype = dtype
        self.engine = Engine(self.plan, max_batch=max_batch, device=device)
        self.feature_dim = 2048

    def extract_batch(self, x):
        return self.engine.forward(x, (OUT_FEATURES,))["features"]

    def close_session(self):
        self.engine.close()
