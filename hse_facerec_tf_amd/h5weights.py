"""Keras ``.h5`` weight files without an HDF5 library (the reference loads ``models/vgg2_mobilenet.h5`` with
``model.load_weights`` at facerec_test.py:322-334, and h5py is not installable on this image).

Two parts:

* ``read_h5(path)`` -- a reader for the subset of HDF5 that h5py / libhdf5 write with their default settings, which is what
  ``keras.Model.save_weights`` / ``save`` produce: version-0 superblock, old-style groups (symbol-table message -> v1 B-tree ->
  symbol-table nodes -> local heap), version-1 object headers with continuation blocks, CONTIGUOUS or compact datasets of
  little-endian floats / integers / fixed-length strings, and attributes stored in the header (``layer_names``,
  ``weight_names``: arrays of fixed-length strings).  Chunked / compressed datasets, the "latest" file format (v2 object
  headers, link messages, dense attributes) and variable-length attribute values are reported as such, not guessed.
* ``keras_mobilenet_graph(path, size)`` -- the weights of Keras' MobileNet-v1 (alpha 1, ``include_top=False``: conv1, conv1_bn,
  conv_dw_k / conv_pw_k and their BatchNormalization layers, k = 1..13) turned into the frozen graph the reference would have
  got from the same model (GlobalAveragePooling2D + ``Reshape((1, 1, 1024), name='reshape_1')``, facerec_test.py:325-333):
  Conv2D / DepthwiseConv2dNative -> FusedBatchNorm(epsilon 1e-3) -> Relu6, input ``input_1``, output ``reshape_1/Reshape`` --
  which ``lower_graph`` then treats like any other frozen graph (every fused kernel applies).
"""
from __future__ import annotations

import struct
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from .graphdef import Graph, GraphNode

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5FormatError(ValueError):
    pass


class _H5:
    def __init__(self, data: bytes):
        self.b = data
        if data[:8] != _SIG:
            raise H5FormatError("not an HDF5 file (no signature at offset 0)")
        ver = data[8]
        if ver not in (0, 1):
            raise H5FormatError("HDF5 superblock version %d: only the classic format h5py writes by default (version 0 / 1) is "
                                "supported -- re-save the weights without libver='latest'" % ver)
        if data[13] != 8 or data[14] != 8:
            raise H5FormatError("HDF5 offsets / lengths of %d / %d bytes (expected 8 / 8)" % (data[13], data[14]))
        pos = 24 + (4 if ver == 1 else 0)
        self.base, _, _, _ = struct.unpack_from("<QQQQ", data, pos)
        pos += 32
        # root group symbol table entry
        _, self.root_header, cache, _ = struct.unpack_from("<QQII", data, pos)
        self.root_scratch = struct.unpack_from("<QQ", data, pos + 24) if cache == 1 else None

    # ---- low level -------------------------------------------------------------------------------------------------
    def u(self, fmt: str, off: int):
        return struct.unpack_from("<" + fmt, self.b, off)

    def messages(self, addr: int) -> List[Tuple[int, int, int]]:
        """(type, offset of the message body, size) of every message of the version-1 object header at ``addr``."""
        addr += self.base
        if self.b[addr:addr + 4] == b"OHDR":
            raise H5FormatError("version-2 object header (file written with libver='latest'): not supported")
        ver, _, nmsg, _, hsize = self.u("BBHII", addr)
        if ver != 1:
            raise H5FormatError("object header version %d at %d" % (ver, addr))
        out = []
        blocks = [(addr + 16, hsize)]
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = self.u("HHB", pos)
                body = pos + 8
                if mtype == 0x10:                                  # continuation: more messages elsewhere
                    o, ln = self.u("QQ", body)
                    blocks.append((o + self.base, ln))
                out.append((mtype, body, msize))
                pos = body + ((msize + 7) & ~7)
        return out

    def heap_string(self, heap_addr: int, off: int) -> str:
        heap_addr += self.base
        if self.b[heap_addr:heap_addr + 4] != b"HEAP":
            raise H5FormatError("no local heap at %d" % heap_addr)
        seg = self.u("Q", heap_addr + 24)[0] + self.base
        end = self.b.index(b"\0", seg + off)
        return self.b[seg + off:end].decode("utf-8")

    def group_entries(self, btree: int, heap: int) -> List[Tuple[str, int]]:
        """(name, object header address) of every link of an old-style group."""
        out = []
        stack = [btree]
        while stack:
            a = stack.pop() + self.base
            if self.b[a:a + 4] == b"TREE":
                ntype, _level, used = self.u("BBH", a + 4)
                if ntype != 0:
                    raise H5FormatError("B-tree node of type %d in a group" % ntype)
                pos = a + 24
                kids = []
                for i in range(used):
                    kids.append(self.u("Q", pos + 8 + 16 * i)[0])
                stack.extend(reversed(kids))
            elif self.b[a:a + 4] == b"SNOD":
                n = self.u("H", a + 6)[0]
                for i in range(n):
                    e = a + 8 + 40 * i
                    name_off, hdr = self.u("QQ", e)
                    out.append((self.heap_string(heap, name_off), hdr))
            else:
                raise H5FormatError("neither a B-tree nor a symbol-table node at %d" % a)
        return out

    # ---- messages ---------------------------------------------------------------------------------------------------
    def dataspace(self, off: int) -> Tuple[int, ...]:
        ver, rank, flags = self.u("BBB", off)
        pos = off + (8 if ver == 1 else 4)
        if ver not in (1, 2):
            raise H5FormatError("dataspace message version %d" % ver)
        return tuple(self.u("Q", pos + 8 * i)[0] for i in range(rank))

    def datatype(self, off: int):
        """-> (numpy dtype or ('S', n) handled as dtype, size of the message)."""
        cv, b0, _b1, _b2, size = self.u("BBBBI", off)
        cls = cv & 15
        if cls == 1:                                              # floating point
            if b0 & 1:
                raise H5FormatError("big-endian floats")
            return np.dtype("<f%d" % size)
        if cls == 0:                                              # fixed point
            if b0 & 1:
                raise H5FormatError("big-endian integers")
            return np.dtype("<%s%d" % ("i" if b0 & 8 else "u", size))
        if cls == 3:                                              # fixed-length string
            return np.dtype("S%d" % size)
        if cls == 9:
            raise H5FormatError("variable-length datatype")
        raise H5FormatError("datatype class %d" % cls)

    def read_attribute(self, off: int) -> Tuple[str, Any]:
        ver = self.b[off]
        if ver == 1:
            _, _, nsize, tsize, ssize = self.u("BBHHH", off)
            pos = off + 8
            pad = lambda n: (n + 7) & ~7
        elif ver in (2, 3):
            _, _, nsize, tsize, ssize = self.u("BBHHH", off)
            pos = off + 8 + (1 if ver == 3 else 0)
            pad = lambda n: n
        else:
            raise H5FormatError("attribute message version %d" % ver)
        name = self.b[pos:pos + nsize].split(b"\0")[0].decode("utf-8")
        pos += pad(nsize)
        try:
            dt = self.datatype(pos)
        except H5FormatError:
            return name, None                                     # e.g. variable-length strings (keras_version, backend): not needed
        pos += pad(tsize)
        shape = self.dataspace(pos)
        pos += pad(ssize)
        n = int(np.prod(shape)) if shape else 1
        arr = np.frombuffer(self.b, dtype=dt, count=n, offset=pos).reshape(shape)
        return name, arr

    def read_dataset(self, msgs) -> Optional[np.ndarray]:
        shape = dt = None
        layout = None
        for mtype, off, size in msgs:
            if mtype == 1:
                shape = self.dataspace(off)
            elif mtype == 3:
                dt = self.datatype(off)
            elif mtype == 8:
                ver = self.b[off]
                if ver == 3:
                    cls = self.b[off + 1]
                    if cls == 1:
                        layout = ("contiguous",) + self.u("QQ", off + 2)
                    elif cls == 0:
                        layout = ("compact", off + 4, self.u("H", off + 2)[0])
                    else:
                        raise H5FormatError("chunked dataset (written with compression or chunks=...): re-save the weights without it")
                elif ver in (1, 2):
                    rank, cls = self.b[off + 1], self.b[off + 2]
                    if cls != 1:
                        raise H5FormatError("data layout class %d in a version-%d layout message" % (cls, ver))
                    layout = ("contiguous", self.u("Q", off + 8)[0], None)
                    del rank
                else:
                    raise H5FormatError("data layout message version %d" % ver)
            elif mtype == 0xB:
                raise H5FormatError("filtered (compressed) dataset: re-save the weights without compression")
        if shape is None or dt is None or layout is None:
            return None
        n = int(np.prod(shape)) if shape else 1
        if layout[0] == "compact":
            return np.frombuffer(self.b, dtype=dt, count=n, offset=layout[1]).reshape(shape).copy()
        if layout[1] == _UNDEF:
            return np.zeros(shape, dt)                            # never written: the fill value
        return np.frombuffer(self.b, dtype=dt, count=n, offset=layout[1] + self.base).reshape(shape).copy()


def read_h5(path_or_bytes) -> Tuple[Dict[str, np.ndarray], Dict[str, Dict[str, Any]]]:
    """-> (datasets {'/group/.../name': array}, attributes {'/path' ('/' = the root): {name: array or None}})."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        data = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    h = _H5(data)
    datasets: Dict[str, np.ndarray] = {}
    attrs: Dict[str, Dict[str, Any]] = {}
    seen = set()

    def walk(path: str, header: int):
        if header in seen:
            return
        seen.add(header)
        msgs = h.messages(header)
        a = {}
        group = None
        for mtype, off, size in msgs:
            if mtype == 0xC:
                k, v = h.read_attribute(off)
                a[k] = v
            elif mtype == 0x11:
                group = h.u("QQ", off)
            elif mtype in (2, 6):
                raise H5FormatError("new-style group (link messages): file written with libver='latest'")
        if a:
            attrs[path or "/"] = a
        if group is not None:
            for name, hdr in h.group_entries(*group):
                walk(path + "/" + name, hdr)
        else:
            d = h.read_dataset(msgs)
            if d is not None:
                datasets[path] = d

    walk("", h.root_header)
    return datasets, attrs


# ----------------------------------------------------------------------------------------------------------------------
# Keras MobileNet-v1 weights -> the frozen graph of the same model
# ----------------------------------------------------------------------------------------------------------------------
def _find(datasets: Dict[str, np.ndarray], layer: str, weight: str) -> np.ndarray:
    """Keras stores weight w of layer l as /[model_weights/]l/l/w:0 (save_weights) -- match by suffix."""
    want = "/%s/%s:0" % (layer, weight)
    hits = [k for k in datasets if k.endswith(want)]
    if not hits:
        raise KeyError("no weight %r of layer %r in the file (datasets: %d)" % (weight, layer, len(datasets)))
    return np.asarray(datasets[min(hits, key=len)], dtype=np.float32)


def keras_mobilenet_graph(path_or_bytes, size: int = 192, bn_epsilon: float = 1e-3) -> Graph:
    """The graph of facerec_test.py:322-334 from its Keras weight file: input_1 [-1, size, size, 3] -> reshape_1/Reshape
    [-1, 1, 1, 1024].  ``size`` must be a multiple of 32: Keras pads its stride-2 convolutions ((0, 1), (0, 1)) + 'valid', which
    is TensorFlow's SAME only on even maps.

    Padding assumption: a weight file does not record the architecture's padding.  This builds every convolution
    with TensorFlow SAME padding -- what ``keras.applications.MobileNet`` is up to Keras 2.1.5 (``padding='same'`` throughout:
    the shape of the reference's own shipped graph, trained with the same code in 2018: conv1 and every stride-2 depthwise
    there are SAME with no Pad node) and again, on even maps, from keras_applications 1.0.4 on (((0, 1), (0, 1)) + 'valid').
    Keras 2.1.6 / 2.2.0 built stride-2 layers as ``ZeroPadding2D((1, 1))`` + 'valid' (windows start one pixel EARLIER): a
    model trained with those two releases needs its graph exported from Keras (the .pb route), not this importer.
    Attributes the reader cannot decode (h5py >= 3 writes ``layer_names`` / ``weight_names`` as variable-length strings) are
    skipped, not fatal: weights are found by dataset PATH."""
    if size % 32:
        raise ValueError("Keras MobileNet's explicit ((0, 1), (0, 1)) padding equals SAME only on even maps: size %d must be a "
                         "multiple of 32" % size)
    datasets, _ = read_h5(path_or_bytes)
    nodes: List[GraphNode] = []
    consts: Dict[str, np.ndarray] = {}

    def const(name: str, arr: np.ndarray) -> str:
        nodes.append(GraphNode(name, "Const", [], {"dtype": {"type": 1}}))
        consts[name] = np.ascontiguousarray(arr, dtype=np.float32)
        return name

    def ints(v):
        return {"list": {"i": list(v)}}

    nodes.append(GraphNode("input_1", "Placeholder", [], {"dtype": {"type": 1},
                                                         "shape": {"shape": {"dim": [{"size": -1}, {"size": size}, {"size": size}, {"size": 3}]}}}))
    x = "input_1"

    def bn_relu6(layer: str, src: str) -> str:
        ins = [src] + [const("%s/%s" % (layer, w), _find(datasets, layer, w)) for w in ("gamma", "beta", "moving_mean", "moving_variance")]
        nodes.append(GraphNode(layer + "/FusedBatchNorm", "FusedBatchNorm", ins,
                               {"epsilon": {"f": float(bn_epsilon)}, "is_training": {"b": False}, "data_format": {"s": b"NHWC"}}))
        relu = layer.replace("_bn", "_relu")
        nodes.append(GraphNode(relu + "/Relu6", "Relu6", [layer + "/FusedBatchNorm"], {}))
        return relu + "/Relu6"

    k = _find(datasets, "conv1", "kernel")
    if k.shape != (3, 3, 3, 32):
        raise H5FormatError("conv1/kernel has shape %r: not MobileNet-v1 alpha 1" % (k.shape,))
    nodes.append(GraphNode("conv1/convolution", "Conv2D", [x, const("conv1/kernel", k)],
                           {"strides": ints((1, 2, 2, 1)), "padding": {"s": b"SAME"}, "data_format": {"s": b"NHWC"}}))
    x = bn_relu6("conv1_bn", "conv1/convolution")
    strides = {2: 2, 4: 2, 6: 2, 12: 2}
    for i in range(1, 14):
        s = strides.get(i, 1)
        dk = _find(datasets, "conv_dw_%d" % i, "depthwise_kernel")
        nodes.append(GraphNode("conv_dw_%d/depthwise" % i, "DepthwiseConv2dNative", [x, const("conv_dw_%d/depthwise_kernel" % i, dk)],
                               {"strides": ints((1, s, s, 1)), "padding": {"s": b"SAME"}, "data_format": {"s": b"NHWC"}}))
        x = bn_relu6("conv_dw_%d_bn" % i, "conv_dw_%d/depthwise" % i)
        pk = _find(datasets, "conv_pw_%d" % i, "kernel")
        nodes.append(GraphNode("conv_pw_%d/convolution" % i, "Conv2D", [x, const("conv_pw_%d/kernel" % i, pk)],
                               {"strides": ints((1, 1, 1, 1)), "padding": {"s": b"SAME"}, "data_format": {"s": b"NHWC"}}))
        x = bn_relu6("conv_pw_%d_bn" % i, "conv_pw_%d/convolution" % i)
    nodes.append(GraphNode("global_average_pooling2d_1/Mean/reduction_indices", "Const", [], {"dtype": {"type": 3}}))
    consts["global_average_pooling2d_1/Mean/reduction_indices"] = np.array([1, 2], np.int32)
    nodes.append(GraphNode("global_average_pooling2d_1/Mean", "Mean", [x, "global_average_pooling2d_1/Mean/reduction_indices"],
                           {"keep_dims": {"b": False}}))
    nodes.append(GraphNode("reshape_1/Reshape/shape", "Const", [], {"dtype": {"type": 3}}))
    consts["reshape_1/Reshape/shape"] = np.array([-1, 1, 1, 1024], np.int32)
    nodes.append(GraphNode("reshape_1/Reshape", "Reshape", ["global_average_pooling2d_1/Mean", "reshape_1/Reshape/shape"], {}))
    g = Graph(nodes)
    g._const.update(consts)          # constants live in the Graph's value cache: no TensorProto round trip
    return g
