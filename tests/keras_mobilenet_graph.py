"""A frozen graph of the SHAPE of the reference's missing `vgg2_mobilenet.pb` (facerec_test.py:212) -- test infrastructure.

That file is Keras MobileNet-v1 (alpha 1, include_top=False) + GlobalAveragePooling2D + Reshape((1,1,1024), name='reshape_1')
(facerec_test.py:322-334, facerec_keras_train.py:52-57) frozen WITHOUT folding: every convolution is followed by an un-folded
BatchNormalization whose inference branch sits behind a `keras_learning_phase` Switch/Merge, activations are Relu6 ops, the
input is `input_1`, the output `reshape_1/Reshape`, and `conv1_bn/keras_learning_phase` must be fed 0 (facerec_test.py:64,118).

The weights here are the SHIPPED trunk (age_gender_tf2_new-01-0.14-0.92_quantized.pb) un-folded: for every layer a random
(moving_mean, moving_variance) is drawn and (gamma, beta) solved so that the BN reproduces the shipped scale / shift; the
convolution kernels are divided by the scale the shipped graph had folded into them.  So this graph computes the same function
as the shipped trunk (to round-off), through the op patterns of the missing file.
"""
import numpy as np

import gb
from hse_facerec_tf_amd import graphdef, lowering

EPS = 1e-3       # Keras MobileNet BatchNormalization epsilon


def _bn(b, prefix, x, scale, shift, rs):
    """Keras BN in inference form behind the learning-phase switch: returns the name of the Merge output."""
    c = scale.shape[0]
    var = rs.uniform(0.5, 2.0, c).astype(np.float32)
    mean = rs.randn(c).astype(np.float32)
    gamma = (scale.astype(np.float64) * np.sqrt(var.astype(np.float64) + EPS)).astype(np.float32)
    beta = (shift.astype(np.float64) + mean.astype(np.float64) * scale.astype(np.float64)).astype(np.float32)
    for nm, v in (("gamma", gamma), ("beta", beta), ("moving_mean", mean), ("moving_variance", var)):
        b.const(prefix + "/" + nm, v)
    b.const(prefix + "/batchnorm/add/y", np.float32(EPS))
    b.node(prefix + "/cond/Switch_1", "Switch", [x, "conv1_bn/keras_learning_phase"])
    b.node(prefix + "/batchnorm/add", "Add", [prefix + "/moving_variance", prefix + "/batchnorm/add/y"])
    b.node(prefix + "/batchnorm/Rsqrt", "Rsqrt", [prefix + "/batchnorm/add"])
    b.node(prefix + "/batchnorm/mul", "Mul", [prefix + "/batchnorm/Rsqrt", prefix + "/gamma"])
    b.node(prefix + "/batchnorm/mul_1", "Mul", [prefix + "/cond/Switch_1", prefix + "/batchnorm/mul"])
    b.node(prefix + "/batchnorm/mul_2", "Mul", [prefix + "/moving_mean", prefix + "/batchnorm/mul"])
    b.node(prefix + "/batchnorm/sub", "Sub", [prefix + "/beta", prefix + "/batchnorm/mul_2"])
    b.node(prefix + "/batchnorm/add_1", "Add", [prefix + "/batchnorm/mul_1", prefix + "/batchnorm/sub"])
    b.node(prefix + "/cond/train_branch", "Neg", [prefix + "/cond/Switch_1:1"])      # hangs off port 1: must be dead
    b.node(prefix + "/cond/Merge", "Merge", [prefix + "/batchnorm/add_1", prefix + "/cond/train_branch"])
    return prefix + "/cond/Merge"


def build(shipped_pb: str, size: int = 192, seed: int = 212) -> bytes:
    """Serialized GraphDef: input_1 [-1,size,size,3] -> reshape_1/Reshape [-1,1,1,1024]."""
    rs = np.random.RandomState(seed)
    g = graphdef.read_graph(shipped_pb)
    plan = lowering.lower_graph(g, "input_1:0", {0: "global_pooling/Mean:0"}, (size, size), fuse=False, pw_math="f32", presplit="none")
    convs = [L for L in plan.layers if L.kind in (lowering.OP_CONV_C3, lowering.OP_DWCONV3X3, lowering.OP_PWCONV_F32)]
    assert len(convs) == 27
    b = gb.GraphBuilder()
    b.placeholder("input_1", [-1, size, size, 3])
    b.placeholder("conv1_bn/keras_learning_phase", None, dtype=10)
    x = "input_1"
    for i, L in enumerate(convs):
        if i == 0:
            name, bn = "conv1", "conv1_bn"
        else:
            blk = (i + 1) // 2
            name, bn = ("conv_dw_%d" % blk, "conv_dw_%d_bn" % blk) if L.kind == lowering.OP_DWCONV3X3 else ("conv_pw_%d" % blk, "conv_pw_%d_bn" % blk)
        cout = L.out_shape[2]
        shift = L.shift if L.shift is not None else np.zeros(cout, np.float32)
        if L.kind == lowering.OP_DWCONV3X3:
            kern = L.w.reshape(3, 3, cout, 1).astype(np.float32)          # the shipped graph keeps the depthwise scale apart
            scale = L.scale.astype(np.float32)
            b.const(name + "/depthwise_kernel", kern)
            b.node(name + "/depthwise", "DepthwiseConv2dNative", [x, name + "/depthwise_kernel"], strides=[1, L.stride, L.stride, 1],
                   padding="SAME", data_format="NHWC")
            x = name + "/depthwise"
        else:
            # the shipped kernel has the BN scale folded in: take a random positive scale out again
            scale = rs.uniform(0.5, 2.0, cout).astype(np.float32)
            kern = (L.w.astype(np.float64) / scale.astype(np.float64)).astype(np.float32)
            b.const(name + "/kernel", kern)
            b.node(name + "/convolution", "Conv2D", [x, name + "/kernel"], strides=[1, L.stride, L.stride, 1], padding="SAME",
                   data_format="NHWC")
            x = name + "/convolution"
        x = _bn(b, bn, x, scale, shift.astype(np.float32), rs)
        relu = ("conv1_relu" if i == 0 else name + "_relu") + "/Relu6"
        b.node(relu, "Relu6", [x])
        x = relu
    b.const("global_average_pooling2d_1/Mean/reduction_indices", np.array([1, 2], np.int32))
    b.node("global_average_pooling2d_1/Mean", "Mean", [x, "global_average_pooling2d_1/Mean/reduction_indices"])
    b.const("reshape_1/Reshape/shape", np.array([-1, 1, 1, 1024], np.int32))
    b.node("reshape_1/Reshape", "Reshape", ["global_average_pooling2d_1/Mean", "reshape_1/Reshape/shape"])
    return b.serialize()
