"""CPU suite: Keras .h5 weight files without an HDF5 library (hse_facerec_tf_amd/h5weights.py; facerec_test.py:322-334 loads
models/vgg2_mobilenet.h5 with model.load_weights).  The files come from tests/h5_writer.py, a hand-rolled writer of the classic
HDF5 layout h5py produces (h5py itself is not installable here): the reader is checked structure by structure, then the
MobileNet importer end to end -- the shipped trunk's weights un-folded into Keras layers, written as save_weights would, read
back, lowered, and run through the plan reference against the shipped graph itself."""
import numpy as np
import pytest

import h5_writer
import plan_ref
from conftest import MODEL_PB
from hse_facerec_tf_amd import graphdef, h5weights, lowering
from oracle import tf_graph as tfo

EPS = 1e-3


def test_reader_round_trip_groups_datasets_attributes():
    rs = np.random.RandomState(0)
    tree = {"a": {"a": {"kernel:0": rs.randn(3, 3, 3, 8).astype(np.float32), "bias:0": rs.randn(8).astype(np.float32)}},
            "empty_layer": {},
            "scalars": {"f64": np.array(3.5, np.float64), "i32": np.arange(6, dtype=np.int32).reshape(2, 3), "nothing": np.zeros((0, 4), np.float32)}}
    for i in range(40):                                      # more than 8 links: several symbol-table nodes under one B-tree node
        tree["layer_%02d" % i] = {"w:0": np.full((2, 2), i, np.float32)}
    attrs = {"/": {"layer_names": np.array([b"a", b"empty_layer", b"scalars"]), "backend": np.array(b"tensorflow")},
             "/a": {"weight_names": np.array([b"a/kernel:0", b"a/bias:0"])}, "/a/a/kernel:0": {"note": np.array([1.5, 2.5], np.float32)}}
    data = h5_writer.write_tree(tree, attrs)
    assert data[:8] == b"\x89HDF\r\n\x1a\n"
    ds, at = h5weights.read_h5(data)
    assert np.array_equal(ds["/a/a/kernel:0"], tree["a"]["a"]["kernel:0"]) and ds["/a/a/kernel:0"].dtype == np.float32
    assert np.array_equal(ds["/a/a/bias:0"], tree["a"]["a"]["bias:0"])
    assert ds["/scalars/f64"].shape == () and float(ds["/scalars/f64"]) == 3.5 and ds["/scalars/f64"].dtype == np.float64
    assert np.array_equal(ds["/scalars/i32"], tree["scalars"]["i32"]) and ds["/scalars/nothing"].shape == (0, 4)
    assert all(np.array_equal(ds["/layer_%02d/w:0" % i], np.full((2, 2), i, np.float32)) for i in range(40))
    assert len(ds) == 2 + 3 + 40
    assert [s.decode() for s in at["/"]["layer_names"]] == ["a", "empty_layer", "scalars"] and at["/"]["backend"].item() == b"tensorflow"
    assert [s.decode() for s in at["/a"]["weight_names"]] == ["a/kernel:0", "a/bias:0"]
    assert np.array_equal(at["/a/a/kernel:0"]["note"], np.array([1.5, 2.5], np.float32))


def test_reader_refuses_what_it_does_not_understand():
    with pytest.raises(h5weights.H5FormatError, match="not an HDF5 file"):
        h5weights.read_h5(b"PK\x03\x04" + b"\0" * 200)
    data = bytearray(h5_writer.write_tree({"w": np.ones(3, np.float32)}))
    v2 = bytearray(data)
    v2[8] = 2                                                # superblock version of libver='latest'
    with pytest.raises(h5weights.H5FormatError, match="superblock version 2"):
        h5weights.read_h5(bytes(v2))
    # a chunked layout message (class 2) where the contiguous one stood
    i = data.index(bytes([3, 1]) + b"\x60")                  # version 3, class 1, then the data address (0x60 = 96: first allocation)
    data[i + 1] = 2
    with pytest.raises(h5weights.H5FormatError, match="chunked"):
        h5weights.read_h5(bytes(data))
    with pytest.raises(FileNotFoundError):
        h5weights.read_h5("/nonexistent/vgg2_mobilenet.h5")


def _keras_layers_of_the_shipped_trunk(seed=7):
    """The shipped graph's trunk as Keras layers: BN un-folded with random statistics, kernels divided by the folded scale."""
    rs = np.random.RandomState(seed)
    plan = lowering.lower_graph(graphdef.read_graph(MODEL_PB), "input_1:0", {0: "global_pooling/Mean:0"}, (64, 64), fuse=False,
                                pw_math="f32", presplit="none")
    convs = [L for L in plan.layers if L.kind in (lowering.OP_CONV_C3, lowering.OP_DWCONV3X3, lowering.OP_PWCONV_F32)]
    assert len(convs) == 27
    layers = {"input_1": {}}
    for i, L in enumerate(convs):
        blk = (i + 1) // 2
        dw = L.kind == lowering.OP_DWCONV3X3
        name = "conv1" if i == 0 else ("conv_dw_%d" % blk if dw else "conv_pw_%d" % blk)
        cout = L.out_shape[2]
        shift = (L.shift if L.shift is not None else np.zeros(cout)).astype(np.float64)
        if dw:
            scale = L.scale.astype(np.float64)
            layers[name] = {"depthwise_kernel": L.w.reshape(3, 3, cout, 1)}
        else:
            scale = rs.uniform(0.5, 2.0, cout)
            layers[name] = {"kernel": (L.w.astype(np.float64) / scale).astype(np.float32)}
        var, mean = rs.uniform(0.5, 2.0, cout), rs.randn(cout)
        layers[name + "_bn"] = {"gamma": scale * np.sqrt(var + EPS), "beta": shift + mean * scale, "moving_mean": mean, "moving_variance": var}
        layers[name + "_relu"] = {}
    layers["global_average_pooling2d_1"] = {}
    layers["reshape_1"] = {}
    return layers


def test_keras_mobilenet_h5_lowers_to_the_function_of_the_shipped_trunk():
    """facerec_test.py:322-334 without Keras: weights file -> graph (input_1 -> reshape_1/Reshape) -> fused plan.  The plan
    reference (tests/plan_ref.py: the wire format executed in NumPy) must give what the fp64 oracle gives on the shipped graph
    the weights were taken from (the BN un-folding is exact to round-off), and every fused kernel kind must apply."""
    size = 64
    data = h5_writer.keras_save_weights(_keras_layers_of_the_shipped_trunk())
    ds, at = h5weights.read_h5(data)
    assert len(ds) == 1 + 13 * 2 + 27 * 4 and [s.decode() for s in at["/"]["layer_names"]][:3] == ["input_1", "conv1", "conv1_bn"]
    g = h5weights.keras_mobilenet_graph(data, size)
    assert g.placeholder_shape("input_1") == [-1, size, size, 3]
    plan = lowering.lower_graph(g, "input_1:0", {0: "reshape_1/Reshape:0"}, None, input_bound=256.0)
    kinds = [L.kind for L in plan.layers]
    assert kinds[0] == lowering.OP_STEM3_F16S and lowering.OP_DWPW_F16S in kinds        # the fused stem and fused blocks apply
    x = np.random.RandomState(3).uniform(-120, 130, (2, size, size, 3)).astype(np.float32)
    got = plan_ref.run(plan.serialize(), x)["features"]
    want = tfo.GraphOracle(MODEL_PB, np.float64).run("global_pooling/Mean:0", {"input_1:0": x}).reshape(2, -1)
    assert got.shape == want.shape == (2, 1024)
    assert np.abs(got - want).max() / np.abs(want).max() < 2e-5
    with pytest.raises(ValueError, match="multiple of 32"):
        h5weights.keras_mobilenet_graph(data, 100)
    del ds["/conv_pw_7_bn/conv_pw_7_bn/beta:0"]
    with pytest.raises(KeyError, match="conv_pw_7_bn"):
        h5weights._find(ds, "conv_pw_7_bn", "beta")
