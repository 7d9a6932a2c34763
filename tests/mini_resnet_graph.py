"""A small ResNet-style frozen GraphDef of the shape a Caffe->TF conversion of `resnet50_ft` takes (the reference's
vgg2_resnet.pb, facerec_test.py:213, is a missing blob): explicit Pad + VALID 7x7/2 stem, BatchNorm as FusedBatchNorm
or as Mul/Add pairs, BiasAdd, 3x3/2 max-pool (SAME or Pad+VALID), bottlenecks with projection / identity shortcuts and
the stride on the first 1x1, global AvgPool named like the reference's output tensor.  Test infrastructure."""
import numpy as np

import gb


def build(seed=0, hw=40, pool="SAME", bn="fused", width=64, head="avgpool"):
    rs = np.random.RandomState(seed)
    b = gb.GraphBuilder()
    b.placeholder("input", [-1, hw, hw, 3])

    def conv(name, src, k, cin, cout, stride, padding):
        b.const(name + "/weights", (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32))
        b.node(name, "Conv2D", [src, name + "/weights"], strides=[1, stride, stride, 1], padding=padding, data_format="NHWC")
        return name

    def batchnorm(name, src, c, damp=1.0):
        g = (damp * rs.uniform(0.8, 1.2, c)).astype(np.float32)
        be = (0.1 * rs.randn(c)).astype(np.float32)
        mu = (0.1 * rs.randn(c)).astype(np.float32)
        var = rs.uniform(0.5, 1.5, c).astype(np.float32)
        if bn == "fused":
            for nm, v in (("gamma", g), ("beta", be), ("mean", mu), ("var", var)):
                b.const("%s/%s" % (name, nm), v)
            b.node(name, "FusedBatchNorm", [src] + ["%s/%s" % (name, nm) for nm in ("gamma", "beta", "mean", "var")],
                   epsilon=1e-5, is_training=False, data_format="NHWC")
        else:   # Caffe BatchNorm + Scale converted to a Mul and an Add
            k = g / np.sqrt(var + 1e-5)
            b.const(name + "/k", k.astype(np.float32))
            b.const(name + "/b", (be - mu * k).astype(np.float32))
            b.node(name + "/mul", "Mul", [src, name + "/k"])
            b.node(name, "Add", [name + "/mul", name + "/b"])
        return name

    def relu(name, src):
        b.node(name, "Relu", [src])
        return name

    b.const("conv1/pad/paddings", np.array([[0, 0], [3, 3], [3, 3], [0, 0]], np.int32))
    b.node("conv1/pad", "Pad", ["input", "conv1/pad/paddings"])
    x = relu("conv1/relu", batchnorm("conv1/bn", conv("conv1/7x7_s2", "conv1/pad", 7, 3, 64, 2, "VALID"), 64))
    if pool == "SAME":
        b.node("pool1/3x3_s2", "MaxPool", [x], ksize=[1, 3, 3, 1], strides=[1, 2, 2, 1], padding="SAME", data_format="NHWC")
    else:       # Caffe ceil-mode pool converted as bottom/right zero pad + VALID
        b.const("pool1/pad/paddings", np.array([[0, 0], [0, 1], [0, 1], [0, 0]], np.int32))
        b.node("pool1/pad", "Pad", [x, "pool1/pad/paddings"])
        b.node("pool1/3x3_s2", "MaxPool", ["pool1/pad"], ksize=[1, 3, 3, 1], strides=[1, 2, 2, 1], padding="VALID", data_format="NHWC")
    x, cin = "pool1/3x3_s2", 64

    def bottleneck(pre, x, cin, mid, cout, stride, proj):
        r = relu(pre + "_1x1_reduce/relu", batchnorm(pre + "_1x1_reduce/bn", conv(pre + "_1x1_reduce", x, 1, cin, mid, stride, "VALID"), mid))
        t = relu(pre + "_3x3/relu", batchnorm(pre + "_3x3/bn", conv(pre + "_3x3", r, 3, mid, mid, 1, "SAME"), mid))
        inc = batchnorm(pre + "_1x1_increase/bn", conv(pre + "_1x1_increase", t, 1, mid, cout, 1, "VALID"), cout, damp=0.4)
        sc = batchnorm(pre + "_1x1_proj/bn", conv(pre + "_1x1_proj", x, 1, cin, cout, stride, "VALID"), cout) if proj else x
        b.node(pre, "Add", [inc, sc])
        return relu(pre + "/relu", pre)

    x = bottleneck("conv2_1", x, 64, width, 4 * width, 1, True)
    x = bottleneck("conv2_2", x, 4 * width, width, 4 * width, 1, False)
    x = bottleneck("conv3_1", x, 4 * width, 2 * width, 8 * width, 2, True)
    final_hw = {"SAME": -(-(-(-hw // 2)) // 2), "PADVALID": ((-(-hw // 2)) + 1 - 3) // 2 + 1}[pool]
    final_hw = -(-final_hw // 2)
    if head == "avgpool":
        b.node("pool5_7x7_s1", "AvgPool", [x], ksize=[1, final_hw, final_hw, 1], strides=[1, 1, 1, 1], padding="VALID", data_format="NHWC")
    else:
        b.const("pool5/axes", np.array([1, 2], np.int32))
        b.node("pool5_7x7_s1", "Mean", [x, "pool5/axes"])
    return b.serialize(), 8 * width
