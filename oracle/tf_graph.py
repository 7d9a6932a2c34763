"""Oracle part 1: TensorFlow-free frozen-GraphDef reader + NumPy graph interpreter.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED vs TensorFlow.

What it restates
----------------
The reference executes frozen graphs with ``tf.import_graph_def`` + ``tf.Session.run``
(facerec_test.py:41-48,58,117-120; facial_analysis.py:55-58,109).  TensorFlow 1.x is
a third-party dependency that is not vendored in /root/reference (no requirements
file pins it; the TF-1 API use implies 1.8-1.13).  This module therefore

  * decodes the GraphDef protobuf *wire format* directly (field numbers from
    TensorFlow's public graph.proto / node_def.proto / attr_value.proto /
    tensor.proto / tensor_shape.proto / types.proto), and
  * evaluates the graph node by node, one NumPy function per TensorFlow op, following
    TensorFlow's published op definitions:
      - SAME padding: out = ceil(in/stride), pad_total = max((out-1)*stride + k - in, 0),
        pad_before = pad_total // 2 (extra pixel goes bottom/right);
      - Dequantize(mode=MIN_FIRST): tensorflow/core/kernels/quantization_utils.h
        ``QuantizedToFloatStruct`` -- scale = (max-min)/255 held in float32,
        min_rounded = round(min/scale)*scale, value = q*scale + min_rounded;
      - Mean = sum/count; Softmax is max-subtracted; Relu/Minimum/Maximum elementwise.

The interpreter is deliberately *unfused and generic*: it does not pattern-match
layers, fold BatchNorm or know what a MobileNet is.  The product
(hse_facerec_tf_amd) compiles the same file into a fused layer plan with its own,
separately written reader; agreement between the two is what the parity tests check.
"""
from __future__ import annotations

import math
import struct
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------- #
# protobuf wire format
# --------------------------------------------------------------------------- #

_DT_NUMPY = {
    1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8,
    9: np.int64, 10: np.bool_, 11: np.int8, 12: np.uint8, 13: np.int32,
}
DT_FLOAT, DT_INT32, DT_QUINT8 = 1, 3, 12


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not (b & 0x80):
            return result, pos
        shift += 7


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _fields(buf: bytes):
    """Yield (field_number, wire_type, value) for one message body."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        fnum, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:  # groups are not used by TensorFlow protos
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fnum, wt, val


def _parse_shape(buf: bytes) -> Optional[List[int]]:
    dims: List[int] = []
    unknown_rank = False
    for f, wt, v in _fields(buf):
        if f == 2:  # Dim
            size = 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    size = _signed64(v2)
            dims.append(size)
        elif f == 3:
            unknown_rank = bool(v)
    return None if unknown_rank else dims


def _packed_or_single(wt: int, v, fmt: str, size: int) -> List:
    if wt == 2:
        return list(struct.unpack("<%d%s" % (len(v) // size, fmt), v))
    return [struct.unpack("<" + fmt, v)[0]]


def _parse_tensor(buf: bytes) -> np.ndarray:
    dtype = 0
    shape: List[int] = []
    content = None
    float_val: List[float] = []
    int_val: List[int] = []
    int64_val: List[int] = []
    double_val: List[float] = []
    bool_val: List[int] = []
    for f, wt, v in _fields(buf):
        if f == 1:
            dtype = v
        elif f == 2:
            shape = _parse_shape(v) or []
        elif f == 4:
            content = v
        elif f == 5:
            float_val += _packed_or_single(wt, v, "f", 4)
        elif f == 6:
            double_val += _packed_or_single(wt, v, "d", 8)
        elif f in (7, 10, 11):
            tgt = {7: int_val, 10: int64_val, 11: bool_val}[f]
            if wt == 2:
                p = 0
                while p < len(v):
                    x, p = _varint(v, p)
                    tgt.append(_signed64(x))
            else:
                tgt.append(_signed64(v))
    npdt = _DT_NUMPY.get(dtype)
    if npdt is None:
        raise ValueError("unsupported tensor dtype %d" % dtype)
    n = int(np.prod(shape)) if shape else 1
    if content is not None and len(content):
        arr = np.frombuffer(content, dtype=npdt).copy()
    else:
        vals = float_val or double_val or int_val or int64_val or bool_val
        if not vals:
            vals = [0]
        arr = np.array(vals, dtype=npdt)
        if arr.size == 1 and n > 1:  # TensorProto splat rule
            arr = np.full(n, arr[0], dtype=npdt)
    arr = arr.reshape(shape)
    # keep the TF dtype enum on the side (quint8 and uint8 share a numpy dtype)
    return arr


class Attr:
    __slots__ = ("s", "i", "f", "b", "type", "shape", "tensor", "list_i", "list_s", "has")

    def __init__(self):
        self.s = self.i = self.f = self.b = self.type = self.shape = self.tensor = None
        self.list_i: List[int] = []
        self.list_s: List[bytes] = []
        self.has = set()


def _parse_attr(buf: bytes) -> Attr:
    a = Attr()
    for f, wt, v in _fields(buf):
        if f == 1:  # ListValue
            for f2, wt2, v2 in _fields(v):
                if f2 == 3:
                    if wt2 == 2:
                        p = 0
                        while p < len(v2):
                            x, p = _varint(v2, p)
                            a.list_i.append(_signed64(x))
                    else:
                        a.list_i.append(_signed64(v2))
                elif f2 == 2:
                    a.list_s.append(bytes(v2))
            a.has.add("list")
        elif f == 2:
            a.s = bytes(v)
        elif f == 3:
            a.i = _signed64(v)
        elif f == 4:
            a.f = struct.unpack("<f", v)[0]
        elif f == 5:
            a.b = bool(v)
        elif f == 6:
            a.type = v
        elif f == 7:
            a.shape = _parse_shape(v)
        elif f == 8:
            a.tensor = _parse_tensor(v)
    return a


class Node:
    __slots__ = ("name", "op", "inputs", "attr")

    def __init__(self, name: str, op: str, inputs: List[str], attr: Dict[str, Attr]):
        self.name, self.op, self.inputs, self.attr = name, op, inputs, attr

    def __repr__(self):
        return "Node(%r, %r, %r)" % (self.name, self.op, self.inputs)


def parse_graphdef(data: bytes) -> List[Node]:
    """GraphDef.node (field 1) -> list of Node, in file order (not topological)."""
    nodes: List[Node] = []
    for f, wt, v in _fields(data):
        if f != 1:
            continue
        name = op = ""
        inputs: List[str] = []
        attr: Dict[str, Attr] = {}
        for f2, wt2, v2 in _fields(v):
            if f2 == 1:
                name = v2.decode("utf-8")
            elif f2 == 2:
                op = v2.decode("utf-8")
            elif f2 == 3:
                inputs.append(v2.decode("utf-8"))
            elif f2 == 5:
                key, val = None, None
                for f3, _, v3 in _fields(v2):
                    if f3 == 1:
                        key = v3.decode("utf-8")
                    elif f3 == 2:
                        val = _parse_attr(v3)
                attr[key] = val if val is not None else Attr()
        nodes.append(Node(name, op, inputs, attr))
    return nodes


def load_graphdef(path: str) -> List[Node]:
    with open(path, "rb") as f:
        return parse_graphdef(f.read())


# --------------------------------------------------------------------------- #
# op restatements (NHWC)
# --------------------------------------------------------------------------- #

def same_pad(in_size: int, k: int, stride: int, dilation: int = 1) -> Tuple[int, int, int]:
    """TensorFlow SAME: returns (out, pad_before, pad_after)."""
    keff = (k - 1) * dilation + 1
    out = -(-in_size // stride)
    total = max((out - 1) * stride + keff - in_size, 0)
    before = total // 2
    return out, before, total - before


def _pads(padding: str, h: int, w: int, kh: int, kw: int, sh: int, sw: int):
    if padding == "SAME":
        oh, pt, pb = same_pad(h, kh, sh)
        ow, pl, pr = same_pad(w, kw, sw)
    elif padding == "VALID":
        oh, ow = (h - kh) // sh + 1, (w - kw) // sw + 1
        pt = pb = pl = pr = 0
    else:
        raise ValueError(padding)
    return oh, ow, pt, pb, pl, pr


def conv2d(x: np.ndarray, k: np.ndarray, strides: Sequence[int], padding: str,
           explicit_pads: Optional[Sequence[int]] = None) -> np.ndarray:
    """tf.nn.conv2d, NHWC x [N,H,W,Cin], HWIO k [kh,kw,Cin,Cout].  Computes in x.dtype."""
    n, h, w, cin = x.shape
    kh, kw, kcin, cout = k.shape
    assert kcin == cin
    sh, sw = strides
    if explicit_pads is not None:
        pt, pb, pl, pr = explicit_pads
        oh = (h + pt + pb - kh) // sh + 1
        ow = (w + pl + pr - kw) // sw + 1
    else:
        oh, ow, pt, pb, pl, pr = _pads(padding, h, w, kh, kw, sh, sw)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((n, oh, ow, cout), dtype=x.dtype)
    k = k.astype(x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            patch = xp[:, dy:dy + (oh - 1) * sh + 1:sh, dx:dx + (ow - 1) * sw + 1:sw, :]
            out += patch.reshape(-1, cin).dot(k[dy, dx]).reshape(n, oh, ow, cout)
    return out


def depthwise_conv2d(x: np.ndarray, k: np.ndarray, strides: Sequence[int], padding: str) -> np.ndarray:
    """tf.nn.depthwise_conv2d_native, k [kh,kw,C,mult]; out channel = c*mult + m."""
    n, h, w, c = x.shape
    kh, kw, kc, mult = k.shape
    assert kc == c
    sh, sw = strides
    oh, ow, pt, pb, pl, pr = _pads(padding, h, w, kh, kw, sh, sw)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((n, oh, ow, c, mult), dtype=x.dtype)
    k = k.astype(x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            patch = xp[:, dy:dy + (oh - 1) * sh + 1:sh, dx:dx + (ow - 1) * sw + 1:sw, :]
            out += patch[..., None] * k[dy, dx]
    return out.reshape(n, oh, ow, c * mult)


def pool2d(x: np.ndarray, ksize: Sequence[int], strides: Sequence[int], padding: str, kind: str) -> np.ndarray:
    """tf.nn.max_pool / avg_pool (NHWC).  SAME max-pool pads with -inf; SAME avg-pool
    divides by the number of valid (un-padded) taps, as TensorFlow does."""
    n, h, w, c = x.shape
    kh, kw = ksize
    sh, sw = strides
    oh, ow, pt, pb, pl, pr = _pads(padding, h, w, kh, kw, sh, sw)
    fill = -np.inf if kind == "max" else 0.0
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=fill)
    acc = None
    for dy in range(kh):
        for dx in range(kw):
            patch = xp[:, dy:dy + (oh - 1) * sh + 1:sh, dx:dx + (ow - 1) * sw + 1:sw, :]
            if acc is None:
                acc = patch.copy()
            elif kind == "max":
                acc = np.maximum(acc, patch)
            else:
                acc = acc + patch
    if kind == "avg":
        ones = np.pad(np.ones((1, h, w, 1), x.dtype), ((0, 0), (pt, pb), (pl, pr), (0, 0)))
        cnt = np.zeros((1, oh, ow, 1), x.dtype)
        for dy in range(kh):
            for dx in range(kw):
                cnt += ones[:, dy:dy + (oh - 1) * sh + 1:sh, dx:dx + (ow - 1) * sw + 1:sw, :]
        acc = acc / cnt
    return acc


def dequantize_min_first(q: np.ndarray, range_min: float, range_max: float) -> np.ndarray:
    """tf.dequantize(mode='MIN_FIRST', T=quint8): quantization_utils.h QuantizedToFloatStruct."""
    rmin = np.float32(range_min)
    rmax = np.float32(range_max)
    if rmin == rmax:
        return np.full(q.shape, rmin, np.float32)
    scale = np.float32(np.float64(np.float32(rmax - rmin)) / 255.0)
    min_rounded = np.float32(np.round(np.float32(rmin / scale)) * scale)
    return (q.astype(np.float32) * scale + min_rounded).astype(np.float32)


def softmax(x: np.ndarray) -> np.ndarray:
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def sigmoid(x: np.ndarray) -> np.ndarray:
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------- #
# interpreter
# --------------------------------------------------------------------------- #

def _split_ref(ref: str) -> Tuple[str, int]:
    if ref.startswith("^"):
        ref = ref[1:]
    if ":" in ref:
        name, idx = ref.rsplit(":", 1)
        return name, int(idx)
    return ref, 0


class GraphOracle:
    """Evaluate tensors of a frozen graph by name, like ``sess.run(fetches, feed_dict)``.

    compute_dtype=np.float64 gives the "truth" used by the parity tests; np.float32
    mimics the reference's fp32 session (summation order still differs from Eigen's,
    which is why the bar is 1e-4 relative, not bit-exact).
    """

    def __init__(self, nodes_or_path, compute_dtype=np.float64):
        nodes = load_graphdef(nodes_or_path) if isinstance(nodes_or_path, str) else nodes_or_path
        self.nodes: Dict[str, Node] = {n.name: n for n in nodes}
        self.order = [n.name for n in nodes]
        self.dt = compute_dtype
        self._const_cache: Dict[str, np.ndarray] = {}

    # -- introspection, as graph.get_tensor_by_name / placeholder shape ---------------
    def tensor_exists(self, ref: str) -> bool:
        name, idx = _split_ref(ref)
        return name in self.nodes and idx == 0

    def placeholder_shape(self, ref: str) -> Optional[List[int]]:
        name, _ = _split_ref(ref)
        a = self.nodes[name].attr.get("shape")
        return None if a is None else a.shape

    # -- evaluation ---------------------------------------------------------------
    def run(self, fetches, feed_dict: Dict[str, np.ndarray]):
        single = isinstance(fetches, str)
        names = [fetches] if single else list(fetches)
        memo: Dict[str, np.ndarray] = {}
        for k, v in feed_dict.items():
            memo[_split_ref(k)[0]] = np.asarray(v)
        outs = [self._eval(_split_ref(f)[0], memo) for f in names]
        return outs[0] if single else outs

    def _eval(self, name: str, memo: Dict[str, np.ndarray]) -> np.ndarray:
        # iterative DFS: the graphs are ~400 nodes deep in a chain
        stack = [name]
        while stack:
            cur = stack[-1]
            if cur in memo:
                stack.pop()
                continue
            node = self.nodes[cur]
            deps = [_split_ref(i)[0] for i in node.inputs if not i.startswith("^")]
            if node.op == "Merge":
                deps = self._merge_live_inputs(node, memo, stack)
                if deps is None:
                    continue
            missing = [d for d in deps if d not in memo]
            if missing:
                stack.extend(missing)
                continue
            memo[cur] = self._apply(node, [memo[d] for d in deps], memo)
            stack.pop()
        return memo[name]

    # Switch/Merge (keras_learning_phase graphs): evaluate the predicate, then only
    # the live branch.  Switch output :0 is the false branch, :1 the true branch.
    def _merge_live_inputs(self, node: Node, memo, stack):
        live = []
        for ref in node.inputs:
            nm, idx = _split_ref(ref)
            src = self._trace_switch(nm, idx)
            if src is None:
                live.append(nm)
                continue
            sw_name, port = src
            pred_name = _split_ref(self.nodes[sw_name].inputs[1])[0]
            if pred_name not in memo:
                stack.append(pred_name)
                return None
            if bool(np.asarray(memo[pred_name]).reshape(-1)[0]) == bool(port):
                live.append(nm)
        assert len(live) >= 1, "Merge %s has no live input" % node.name
        return live[:1]

    def _trace_switch(self, name: str, idx: int):
        """Walk up single-input chains to find the Switch (and port) a tensor hangs off."""
        seen = 0
        while seen < 64:
            node = self.nodes[name]
            if node.op == "Switch":
                return name, idx
            data_in = [i for i in node.inputs if not i.startswith("^")]
            found = None
            for ref in data_in:
                nm, ix = _split_ref(ref)
                if self.nodes[nm].op not in ("Const",):
                    found = (nm, ix)
                    break
            if found is None:
                return None
            name, idx = found
            seen += 1
        return None

    def _apply(self, node: Node, xs: List[np.ndarray], memo) -> np.ndarray:
        op, at = node.op, node.attr
        dt = self.dt
        if op == "Const":
            if node.name not in self._const_cache:
                self._const_cache[node.name] = at["value"].tensor
            return self._const_cache[node.name]
        if op == "Placeholder" or op == "PlaceholderWithDefault":
            if op == "PlaceholderWithDefault":
                return xs[0]
            raise KeyError("placeholder %r was not fed" % node.name)
        if op in ("Identity", "StopGradient"):
            return xs[0]
        if op == "Switch":
            return xs[0]  # routing handled at Merge
        if op == "Merge":
            return xs[0]
        if op == "Dequantize":
            mode = at["mode"].s.decode() if "mode" in at and at["mode"].s else "MIN_COMBINED"
            assert mode == "MIN_FIRST", "only MIN_FIRST is restated (the mode graph_transforms emits)"
            return dequantize_min_first(xs[0], float(xs[1]), float(xs[2]))
        if op == "Conv2D":
            s = at["strides"].list_i
            assert (at["data_format"].s or b"NHWC") == b"NHWC"
            return conv2d(xs[0].astype(dt), xs[1], (s[1], s[2]), at["padding"].s.decode())
        if op == "DepthwiseConv2dNative":
            s = at["strides"].list_i
            return depthwise_conv2d(xs[0].astype(dt), xs[1], (s[1], s[2]), at["padding"].s.decode())
        if op in ("MaxPool", "AvgPool"):
            k, s = at["ksize"].list_i, at["strides"].list_i
            return pool2d(xs[0].astype(dt), (k[1], k[2]), (s[1], s[2]), at["padding"].s.decode(),
                          "max" if op == "MaxPool" else "avg")
        if op == "Pad":
            p = np.asarray(xs[1]).astype(int)
            return np.pad(xs[0], [(int(a), int(b)) for a, b in p])
        if op in ("Add", "AddV2", "BiasAdd"):
            return xs[0].astype(dt) + xs[1].astype(dt)
        if op == "Sub":
            return xs[0].astype(dt) - xs[1].astype(dt)
        if op == "Mul":
            return xs[0].astype(dt) * xs[1].astype(dt)
        if op == "RealDiv":
            return xs[0].astype(dt) / xs[1].astype(dt)
        if op == "Neg":
            return -xs[0]
        if op == "Abs":
            return np.abs(xs[0])
        if op == "Rsqrt":
            return 1.0 / np.sqrt(xs[0].astype(dt))
        if op == "Relu":
            return np.maximum(xs[0], 0)
        if op == "Relu6":
            return np.minimum(np.maximum(xs[0], 0), 6)
        if op == "Minimum":
            return np.minimum(xs[0], xs[1].astype(xs[0].dtype))
        if op == "Maximum":
            return np.maximum(xs[0], xs[1].astype(xs[0].dtype))
        if op == "Mean":
            axes = tuple(int(a) for a in np.asarray(xs[1]).reshape(-1))
            keep = bool(at["keep_dims"].b) if "keep_dims" in at else False
            return xs[0].astype(dt).mean(axis=axes, keepdims=keep)
        if op == "Max":
            axes = tuple(int(a) for a in np.asarray(xs[1]).reshape(-1))
            keep = bool(at["keep_dims"].b) if "keep_dims" in at else False
            return xs[0].max(axis=axes, keepdims=keep)
        if op == "Sum":
            axes = tuple(int(a) for a in np.asarray(xs[1]).reshape(-1))
            keep = bool(at["keep_dims"].b) if "keep_dims" in at else False
            return xs[0].astype(dt).sum(axis=axes, keepdims=keep)
        if op == "Exp":
            return np.exp(xs[0].astype(dt))
        if op == "MatMul":
            a, b = xs[0].astype(dt), xs[1].astype(dt)
            if "transpose_a" in at and at["transpose_a"].b:
                a = a.T
            if "transpose_b" in at and at["transpose_b"].b:
                b = b.T
            return a.dot(b)
        if op == "Softmax":
            return softmax(xs[0].astype(dt))
        if op == "Sigmoid":
            return sigmoid(xs[0].astype(dt))
        if op == "Reshape":
            return xs[0].reshape([int(d) for d in np.asarray(xs[1]).reshape(-1)])
        if op == "Squeeze":
            dims = at["squeeze_dims"].list_i if "squeeze_dims" in at else []
            return np.squeeze(xs[0], axis=tuple(dims) if dims else None)
        if op == "Transpose":
            return np.transpose(xs[0], [int(d) for d in np.asarray(xs[1]).reshape(-1)])
        if op == "Shape":
            return np.array(xs[0].shape, np.int32)
        if op in ("FusedBatchNorm", "FusedBatchNormV3"):
            assert not (at["is_training"].b if "is_training" in at else True), "training-mode BN in a frozen graph"
            eps = at["epsilon"].f if "epsilon" in at and at["epsilon"].f is not None else 1e-3
            x, g, b, mu, var = [a.astype(dt) for a in xs[:5]]
            return (x - mu) / np.sqrt(var + eps) * g + b
        raise NotImplementedError("oracle has no restatement for TensorFlow op %r (node %r)" % (op, node.name))
