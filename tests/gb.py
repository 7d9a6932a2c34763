"""Tiny GraphDef *writer* (protobuf wire format) so tests can build frozen graphs of shapes the
reference's missing files have (un-folded BN, learning-phase Switch/Merge) -- test infrastructure."""
import struct

import numpy as np

_DT = {np.dtype(np.float32): 1, np.dtype(np.int32): 3, np.dtype(np.uint8): 4, np.dtype(np.bool_): 10}


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(f, wt):
    return _varint((f << 3) | wt)


def _ld(f, payload):
    return _key(f, 2) + _varint(len(payload)) + payload


def _shape(dims):
    if dims is None:
        return _key(3, 0) + _varint(1)
    return b"".join(_ld(2, _key(1, 0) + _varint(d)) for d in dims)


class GraphBuilder:
    def __init__(self):
        self.nodes = []

    def _attr(self, key, value_bytes):
        return _ld(5, _ld(1, key.encode()) + _ld(2, value_bytes))

    def placeholder(self, name, shape, dtype=1):
        body = _ld(1, name.encode()) + _ld(2, b"Placeholder")
        body += self._attr("dtype", _key(6, 0) + _varint(dtype))
        body += self._attr("shape", _ld(7, _shape(shape)))
        self.nodes.append(body)

    def const(self, name, arr):
        arr = np.asarray(arr)
        t = _key(1, 0) + _varint(_DT[arr.dtype]) + _ld(2, _shape(list(arr.shape))) + _ld(4, arr.tobytes())
        body = _ld(1, name.encode()) + _ld(2, b"Const")
        body += self._attr("dtype", _key(6, 0) + _varint(_DT[arr.dtype]))
        body += self._attr("value", _ld(8, t))
        self.nodes.append(body)

    def node(self, name, op, inputs, **attrs):
        body = _ld(1, name.encode()) + _ld(2, op.encode())
        for i in inputs:
            body += _ld(3, i.encode())
        for k, v in attrs.items():
            if isinstance(v, str):
                body += self._attr(k, _ld(2, v.encode()))
            elif isinstance(v, bool):
                body += self._attr(k, _key(5, 0) + _varint(int(v)))
            elif isinstance(v, float):
                body += self._attr(k, _key(4, 5) + struct.pack("<f", v))
            elif isinstance(v, (list, tuple)):
                body += self._attr(k, _ld(1, _ld(3, b"".join(_varint(int(x)) for x in v))))
            else:
                raise TypeError(k)
        self.nodes.append(body)

    def serialize(self):
        return b"".join(_ld(1, n) for n in self.nodes)
