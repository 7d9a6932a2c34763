"""TensorFlow-free reader for frozen ``GraphDef`` files (``*.pb``).

Replaces ``tf.GraphDef().ParseFromString`` + ``tf.import_graph_def`` + ``graph.get_tensor_by_name``
(facerec_test.py:41-48,60-70; facial_analysis.py:319-332,84-91) for the purpose of *reading*
the graph; execution is the job of the lowering (lowering.py) and the HIP engine.

Schema-driven protobuf wire decoder: the message layouts below are TensorFlow's public
``graph.proto``, ``node_def.proto``, ``attr_value.proto``, ``tensor.proto``,
``tensor_shape.proto`` and ``types.proto`` field numbers.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Any, Dict, Iterator, List, Optional, Tuple, Sequence

import numpy as np

__all__ = ["Graph", "GraphNode", "read_graph", "DT_FLOAT", "DT_QUINT8"]

# ---- types.proto ---------------------------------------------------------------------
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_INT64, DT_BOOL, DT_QUINT8 = 1, 2, 3, 4, 9, 10, 12
_NP_OF_DT = {DT_FLOAT: "<f4", DT_DOUBLE: "<f8", DT_INT32: "<i4", DT_UINT8: "u1", 5: "<i2", 6: "i1",
             DT_INT64: "<i8", DT_BOOL: "?", 11: "i1", DT_QUINT8: "u1", 13: "<i4"}

# ---- schema: message -> {field: (name, kind, repeated)} ------------------------------
# kinds: 'str', 'bytes', 'int' (varint, signed 64), 'bool', 'f32', 'f64', ('msg', NAME), 'enum'
_SCHEMA: Dict[str, Dict[int, Tuple[str, Any, bool]]] = {
    "GraphDef": {1: ("node", ("msg", "NodeDef"), True)},
    "NodeDef": {1: ("name", "str", False), 2: ("op", "str", False), 3: ("input", "str", True),
                4: ("device", "str", False), 5: ("attr", ("msg", "AttrEntry"), True)},
    "AttrEntry": {1: ("key", "str", False), 2: ("value", ("msg", "AttrValue"), False)},
    "AttrValue": {1: ("list", ("msg", "ListValue"), False), 2: ("s", "bytes", False), 3: ("i", "int", False),
                  4: ("f", "f32", False), 5: ("b", "bool", False), 6: ("type", "int", False),
                  7: ("shape", ("msg", "TensorShape"), False), 8: ("tensor", ("msg", "Tensor"), False)},
    "ListValue": {2: ("s", "bytes", True), 3: ("i", "int", True), 4: ("f", "f32", True), 5: ("b", "bool", True),
                  6: ("type", "int", True)},
    "TensorShape": {2: ("dim", ("msg", "Dim"), True), 3: ("unknown_rank", "bool", False)},
    "Dim": {1: ("size", "int", False), 2: ("name", "str", False)},
    "Tensor": {1: ("dtype", "int", False), 2: ("tensor_shape", ("msg", "TensorShape"), False),
               4: ("tensor_content", "bytes", False), 5: ("float_val", "f32", True),
               6: ("double_val", "f64", True), 7: ("int_val", "int", True), 10: ("int64_val", "int", True),
               11: ("bool_val", "bool", True)},
}


def _read_varint(b: memoryview, i: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        c = b[i]
        i += 1
        out |= (c & 0x7F) << shift
        if c < 0x80:
            return out, i
        shift += 7


def _to_signed(v: int) -> int:
    return v - (1 << 64) if v & (1 << 63) else v


def _scalar(kind, wire: int, raw):
    if kind == "int":
        return _to_signed(raw)
    if kind == "bool":
        return bool(raw)
    if kind == "f32":
        return struct.unpack("<f", raw)[0]
    if kind == "f64":
        return struct.unpack("<d", raw)[0]
    if kind == "str":
        return bytes(raw).decode("utf-8")
    if kind == "bytes":
        return bytes(raw)
    raise ValueError(kind)


def _unpack_packed(kind, raw: memoryview) -> List:
    if kind == "f32":
        return list(struct.unpack("<%df" % (len(raw) // 4), raw))
    if kind == "f64":
        return list(struct.unpack("<%dd" % (len(raw) // 8), raw))
    out, i = [], 0
    while i < len(raw):
        v, i = _read_varint(raw, i)
        out.append(bool(v) if kind == "bool" else _to_signed(v))
    return out


def _decode(msg: str, b: memoryview) -> Dict[str, Any]:
    schema = _SCHEMA[msg]
    out: Dict[str, Any] = {}
    i, n = 0, len(b)
    while i < n:
        key, i = _read_varint(b, i)
        fnum, wire = key >> 3, key & 7
        if wire == 0:
            raw, i = _read_varint(b, i)
        elif wire == 1:
            raw, i = b[i:i + 8], i + 8
        elif wire == 2:
            ln, i = _read_varint(b, i)
            raw, i = b[i:i + ln], i + ln
        elif wire == 5:
            raw, i = b[i:i + 4], i + 4
        else:
            raise ValueError("GraphDef: unsupported wire type %d in %s" % (wire, msg))
        spec = schema.get(fnum)
        if spec is None:
            continue  # unknown field: skip, as protobuf does
        name, kind, repeated = spec
        if isinstance(kind, tuple):
            val = _decode(kind[1], raw)
            if repeated:
                out.setdefault(name, []).append(val)
            else:
                out[name] = val
        elif repeated:
            lst = out.setdefault(name, [])
            if wire == 2 and kind in ("int", "bool", "f32", "f64"):
                lst.extend(_unpack_packed(kind, raw))
            else:
                lst.append(_scalar(kind, wire, raw))
        else:
            out[name] = _scalar(kind, wire, raw)
    return out


def _shape_of(ts: Optional[Dict[str, Any]]) -> Optional[List[int]]:
    if ts is None:
        return []
    if ts.get("unknown_rank"):
        return None
    return [d.get("size", 0) for d in ts.get("dim", [])]


def _tensor_to_numpy(t: Dict[str, Any]) -> np.ndarray:
    dt = t.get("dtype", 0)
    if dt not in _NP_OF_DT:
        raise ValueError("GraphDef: tensor dtype %d is not supported" % dt)
    npdt = np.dtype(_NP_OF_DT[dt])
    shape = _shape_of(t.get("tensor_shape")) or []
    count = int(np.prod(shape, dtype=np.int64)) if shape else 1
    content = t.get("tensor_content")
    if content:
        arr = np.frombuffer(content, dtype=npdt)
    else:
        vals = None
        for k in ("float_val", "double_val", "int_val", "int64_val", "bool_val"):
            if t.get(k):
                vals = t[k]
                break
        arr = np.array(vals if vals is not None else [0], dtype=npdt)
        if arr.size != count:  # a single value fills the tensor; a short list repeats its last element
            full = np.empty(count, dtype=npdt)
            full[:arr.size] = arr[:count]
            full[arr.size:] = arr[-1]
            arr = full
    return arr.reshape(shape)


@dataclass
class GraphNode:
    name: str
    op: str
    inputs: List[str]
    attrs: Dict[str, Dict[str, Any]] = field(default_factory=dict, repr=False)

    def attr_s(self, key: str, default: Optional[str] = None) -> Optional[str]:
        a = self.attrs.get(key)
        return a["s"].decode() if a and "s" in a else default

    def attr_ints(self, key: str) -> List[int]:
        a = self.attrs.get(key)
        return list(a["list"].get("i", [])) if a and "list" in a else []

    def attr_b(self, key: str, default: bool = False) -> bool:
        a = self.attrs.get(key)
        return bool(a["b"]) if a and "b" in a else default

    def attr_f(self, key: str, default: float) -> float:
        a = self.attrs.get(key)
        return float(a["f"]) if a and "f" in a else default

    def attr_type(self, key: str) -> Optional[int]:
        a = self.attrs.get(key)
        return a.get("type") if a else None


def _parse_ref(ref: str) -> Tuple[str, int, bool]:
    ctrl = ref.startswith("^")
    if ctrl:
        ref = ref[1:]
    name, _, idx = ref.partition(":")
    return name, int(idx) if idx else 0, ctrl


class Graph:
    """Name-addressable view of a frozen graph (node order in the file is NOT topological)."""

    def __init__(self, nodes: List[GraphNode]):
        self.nodes = nodes
        self.by_name: Dict[str, GraphNode] = {n.name: n for n in nodes}
        self._const: Dict[str, np.ndarray] = {}
        self._consumers: Optional[Dict[str, List[GraphNode]]] = None
        self._data_inputs: Dict[str, Tuple[GraphNode, List[Tuple[GraphNode, int]]]] = {}

    def __contains__(self, name: str) -> bool:
        return name in self.by_name

    def __iter__(self) -> Iterator[GraphNode]:
        return iter(self.nodes)

    def node(self, name: str) -> GraphNode:
        return self.by_name[name]

    # graph.get_tensor_by_name(name) raises KeyError for unknown names and ValueError for a
    # malformed 'op:idx' string (facerec_test.py:60-64); mirror both.
    def get_tensor_by_name(self, tensor_name: str) -> Tuple[GraphNode, int]:
        if ":" not in tensor_name:
            raise ValueError("The name %r looks like an (invalid) Operation name, not a Tensor. Tensor names "
                             "must be of the form \"<op_name>:<output_index>\"." % tensor_name)
        name, idx, _ = _parse_ref(tensor_name)
        if name not in self.by_name:
            raise KeyError("The name %r refers to a Tensor which does not exist. The operation, %r, does not "
                           "exist in the graph." % (tensor_name, name))
        if idx != 0 and self.by_name[name].op not in ("Switch", "FusedBatchNorm", "FusedBatchNormV3"):
            raise KeyError("The name %r refers to a Tensor which does not exist." % tensor_name)
        return self.by_name[name], idx

    def data_inputs(self, node: GraphNode) -> List[Tuple[GraphNode, int]]:
        """(producer node, output index) of every data input, in order.  Memoised per node (the graph does not change after
        loading; graph walkers such as mtcnn._DeviceNet ask ~450 times per frame) -- treat the list as read-only."""
        cached = self._data_inputs.get(node.name)
        if cached is not None and cached[0] is node:
            return cached[1]
        out = []
        for ref in node.inputs:
            name, idx, ctrl = _parse_ref(ref)
            if not ctrl:
                out.append((self.by_name[name], idx))
        self._data_inputs[node.name] = (node, out)
        return out

    def consumers(self, name: str) -> List[GraphNode]:
        if self._consumers is None:
            cons: Dict[str, List[GraphNode]] = {}
            for n in self.nodes:
                for ref in n.inputs:
                    nm, _, ctrl = _parse_ref(ref)
                    if not ctrl:
                        cons.setdefault(nm, []).append(n)
            self._consumers = cons
        return self._consumers.get(name, [])

    def placeholder_shape(self, name: str) -> Optional[List[int]]:
        a = self.by_name[name].attrs.get("shape")
        if not a or "shape" not in a:
            return None
        return _shape_of(a["shape"])

    def const_value(self, node: GraphNode) -> np.ndarray:
        v = self._const.get(node.name)
        if v is None:
            v = _tensor_to_numpy(node.attrs["value"]["tensor"])
            self._const[node.name] = v
        return v

    def with_prefix(self, prefix: str) -> "Graph":
        """What ``tf.import_graph_def(graph_def, name=prefix)`` does to names (facerec_test.py:46-47,
        facial_analysis.py:328-332): every node becomes ``prefix/<name>`` and every input reference -- data or
        ``^control`` -- follows.  ``''`` (the default of both reference loaders) leaves the graph as it is.  Attributes are
        shared with this graph (constants are not copied)."""
        if not prefix:
            return self
        p = prefix.rstrip("/") + "/"

        def ref(r: str) -> str:
            return "^" + p + r[1:] if r.startswith("^") else p + r
        return Graph([GraphNode(p + n.name, n.op, [ref(r) for r in n.inputs], n.attrs) for n in self.nodes])

    @staticmethod
    def merged(graphs: Sequence["Graph"]) -> "Graph":
        """Several imported graphs as ONE graph, the way facial_analysis.py:55-58 imports mtcnn.pb and the age / gender
        file(s) into one ``full_graph`` behind one session.  A node name that two of them define raises ValueError, as
        ``import_graph_def`` does for a name that is already in use."""
        nodes: List[GraphNode] = []
        seen: Dict[str, int] = {}
        for gi, g in enumerate(graphs):
            for n in g.nodes:
                if n.name in seen:
                    raise ValueError("node name %r is defined by graph %d and graph %d: import them under different prefixes"
                                     % (n.name, seen[n.name], gi))
                seen[n.name] = gi
                nodes.append(n)
        return Graph(nodes)


def read_graph(path_or_bytes) -> Graph:
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        data = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:   # missing file -> FileNotFoundError (TF: NotFoundError)
            data = f.read()
    gd = _decode("GraphDef", memoryview(data))
    nodes = []
    for nd in gd.get("node", []):
        attrs = {e["key"]: e.get("value", {}) for e in nd.get("attr", [])}
        nodes.append(GraphNode(nd.get("name", ""), nd.get("op", ""), list(nd.get("input", [])), attrs))
    if not nodes:
        raise ValueError("no nodes decoded: not a frozen GraphDef")
    return Graph(nodes)
