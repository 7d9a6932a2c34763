"""Hand-rolled HDF5 WRITER -- test infrastructure for hse_facerec_tf_amd/h5weights.py (h5py is not installable on this image).

Writes what libhdf5 / h5py write with default settings, structure for structure: version-0 superblock, old-style groups (symbol
table message -> v1 B-tree node -> symbol-table nodes of at most 8 links, names in a local heap), version-1 object headers,
contiguous datasets with version-1 dataspace / datatype / version-3 layout messages, attributes as version-1 attribute messages
(arrays of fixed-length strings, as Keras stores ``layer_names`` / ``weight_names``).  ``keras_save_weights`` lays a dict of
layer weights out exactly as ``keras.Model.save_weights`` does: /layer/layer/weight:0."""
import struct

import numpy as np

_UNDEF = 0xFFFFFFFFFFFFFFFF


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


class H5Writer:
    def __init__(self):
        self.buf = bytearray(b"\0" * 96)                      # superblock (56 + 40) patched in at the end

    def _alloc(self, data: bytes) -> int:
        while len(self.buf) % 8:
            self.buf.append(0)
        a = len(self.buf)
        self.buf.extend(data)
        return a

    # ---- messages ----------------------------------------------------------------------------------------------------
    @staticmethod
    def _dtype_msg(dt: np.dtype) -> bytes:
        dt = np.dtype(dt)
        if dt.kind == "f":
            assert dt.itemsize in (4, 8)
            if dt.itemsize == 4:
                props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
                return struct.pack("<BBBBI", 0x11, 0x20, 31, 0, 4) + props
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            return struct.pack("<BBBBI", 0x11, 0x20, 63, 0, 8) + props
        if dt.kind in "iu":
            return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
        if dt.kind == "S":
            return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, dt.itemsize)       # null-padded ASCII
        raise TypeError(dt)

    @staticmethod
    def _space_msg(shape) -> bytes:
        return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", d) for d in shape)

    def _attr_msg(self, name: str, value) -> bytes:
        arr = np.asarray(value)
        if arr.dtype.kind == "U":
            arr = np.char.encode(arr, "utf-8")
        nm = name.encode() + b"\0"
        dtm, spm = self._dtype_msg(arr.dtype), self._space_msg(arr.shape)
        return struct.pack("<BBHHH", 1, 0, len(nm), len(dtm), len(spm)) + _pad8(nm) + _pad8(dtm) + _pad8(spm) + arr.tobytes()

    def _object_header(self, msgs) -> int:
        body = b"".join(struct.pack("<HHB3x", t, len(_pad8(m)), 0) + _pad8(m) for t, m in msgs)
        return self._alloc(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)

    # ---- objects -----------------------------------------------------------------------------------------------------
    def dataset(self, arr, attrs=None) -> int:
        arr = np.asarray(arr)
        if not arr.flags.c_contiguous:
            arr = np.ascontiguousarray(arr)
        data = self._alloc(arr.tobytes()) if arr.size else _UNDEF
        msgs = [(1, self._space_msg(arr.shape)), (3, self._dtype_msg(arr.dtype)),
                (5, struct.pack("<BBBB", 2, 2, 0, 0)),                                       # fill value: version 2, never written
                (8, struct.pack("<BBQQ", 3, 1, data, arr.nbytes))]
        msgs += [(0xC, self._attr_msg(k, v)) for k, v in (attrs or {}).items()]
        return self._object_header(msgs)

    def group(self, links, attrs=None) -> int:
        """links: {name: object header address}.  Returns the group's object header address."""
        names = sorted(links)
        heap_data = bytearray(b"\0" * 8)                     # offset 0: the empty string
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data.extend(_pad8(n.encode() + b"\0"))
        free = len(heap_data)
        heap_data.extend(struct.pack("<QQ", 1, 16))          # one free block at the end (next = 1: last; size 16)
        seg = self._alloc(bytes(heap_data))
        heap = self._alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free, seg))
        snods, keys = [], [0]
        for i in range(0, max(len(names), 1), 8):
            chunk = names[i:i + 8]
            ent = b"".join(struct.pack("<QQII16x", offs[n], links[n], 0, 0) for n in chunk)
            snods.append(self._alloc(b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk)) + ent + b"\0" * (40 * (8 - len(chunk)))))
            keys.append(offs[chunk[-1]] if chunk else 0)
        assert len(snods) <= 32, "one B-tree level holds 32 x 8 links"
        kc = b"".join(struct.pack("<QQ", keys[i], snods[i]) for i in range(len(snods))) + struct.pack("<Q", keys[len(snods)])
        tree = self._alloc(b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), _UNDEF, _UNDEF) + kc + b"\0" * (24 + 33 * 8 + 32 * 8 - 24 - len(kc)))
        msgs = [(0x11, struct.pack("<QQ", tree, heap))] + [(0xC, self._attr_msg(k, v)) for k, v in (attrs or {}).items()]
        self._scratch = (tree, heap)
        return self._object_header(msgs)

    def finish(self, root_header: int) -> bytes:
        tree, heap = self._scratch                           # (of the group written last = the root)
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", 4, 16, 0)
        sb += struct.pack("<QQQQ", 0, _UNDEF, len(self.buf), _UNDEF)
        sb += struct.pack("<QQII", 0, root_header, 1, 0) + struct.pack("<QQ", tree, heap)
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def write_tree(tree: dict, attrs: dict = None) -> bytes:
    """tree: nested dicts, leaves = arrays; attrs: {'/path' or '/': {name: value}}."""
    w = H5Writer()
    attrs = attrs or {}

    def rec(node, path):
        if isinstance(node, dict):
            return w.group({k: rec(v, path + "/" + k) for k, v in node.items()}, attrs.get(path or "/"))
        return w.dataset(node, attrs.get(path))
    return w.finish(rec(tree, ""))


def keras_save_weights(layers: dict) -> bytes:
    """layers: {layer name: {weight name: array}} (ordered) -> the bytes keras.Model.save_weights would write."""
    tree, attrs = {}, {"/": {"layer_names": np.array([n.encode() for n in layers]), "backend": np.array(b"tensorflow"),
                             "keras_version": np.array(b"2.2.4")}}
    for lname, ws in layers.items():
        tree[lname] = {lname: {wn + ":0": np.asarray(a, np.float32) for wn, a in ws.items()}} if ws else {}
        attrs["/" + lname] = {"weight_names": np.array([("%s/%s:0" % (lname, wn)).encode() for wn in ws] or [b""])}
    return write_tree(tree, attrs)
