"""Oracle part 3: the same frozen graph lowered node-by-node onto torch-CPU (oneDNN).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED vs TensorFlow.

Two uses:
  * an *independent* convolution implementation (``torch.nn.functional.conv2d``) to
    cross-check the NumPy restatement in oracle/tf_graph.py -- different code computing
    the same TensorFlow-defined arithmetic;
  * the ``cpu_baseline`` leg of bench.py: TensorFlow cannot run on this image, so the
    "reference TF-CPU path" (facerec_test.py:116-120 -- one ``sess.run`` per image) is
    timed as *the reference's own graph executed op-by-op, unfused, in fp32 on the host
    cores by oneDNN*.  It is labelled ``kind: "port"``: a TF-equivalent graph, not TensorFlow.

Tensors are kept NHWC-logical; convs see them as channels_last NCHW views (zero copy).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from .tf_graph import GraphOracle, Node, same_pad, dequantize_min_first


def _t(x, dtype=torch.float32):
    if isinstance(x, torch.Tensor):
        return x
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


class TorchGraphOracle(GraphOracle):
    def __init__(self, nodes_or_path, dtype=torch.float32):
        super().__init__(nodes_or_path, np.float32)
        self.tdt = dtype
        self._wcache: Dict[str, torch.Tensor] = {}

    def run(self, fetches, feed_dict):
        with torch.no_grad():
            feed = {k: _t(v, self.tdt) if np.asarray(v).ndim > 0 else v for k, v in feed_dict.items()}
            out = super().run(fetches, feed)
        if isinstance(out, list):
            return [o.numpy() if isinstance(o, torch.Tensor) else np.asarray(o) for o in out]
        return out.numpy() if isinstance(out, torch.Tensor) else np.asarray(out)

    def _weight(self, node_name: str, arr, kind: str) -> torch.Tensor:
        key = node_name + "/" + kind
        w = self._wcache.get(key)
        if w is None:
            a = arr.numpy() if isinstance(arr, torch.Tensor) else np.asarray(arr)
            if kind == "conv":       # HWIO -> OIHW
                w = _t(a.transpose(3, 2, 0, 1), self.tdt).contiguous(memory_format=torch.channels_last)
            elif kind == "dw":       # [kh,kw,C,1] -> [C,1,kh,kw]
                w = _t(a.transpose(2, 3, 0, 1), self.tdt).contiguous()
            else:
                w = _t(a, self.tdt)
            self._wcache[key] = w
        return w

    def _apply(self, node: Node, xs: List, memo):
        op, at = node.op, node.attr
        if op == "Const":
            return super()._apply(node, xs, memo)
        if op == "Dequantize":
            key = node.name + "/deq"
            if key not in self._wcache:
                q = xs[0].numpy() if isinstance(xs[0], torch.Tensor) else xs[0]
                self._wcache[key] = dequantize_min_first(np.asarray(q), float(xs[1]), float(xs[2]))
            return self._wcache[key]
        if op in ("Conv2D", "DepthwiseConv2dNative"):
            x = _t(xs[0], self.tdt)
            s = at["strides"].list_i
            kshape = xs[1].shape
            kh, kw = int(kshape[0]), int(kshape[1])
            n, h, w_, c = x.shape
            if at["padding"].s == b"SAME":
                _, pt, pb = same_pad(h, kh, s[1])
                _, pl, pr = same_pad(w_, kw, s[2])
            else:
                pt = pb = pl = pr = 0
            xn = x.permute(0, 3, 1, 2)
            if pt or pb or pl or pr:
                xn = F.pad(xn, (pl, pr, pt, pb))
            if op == "Conv2D":
                y = F.conv2d(xn, self._weight(node.inputs[1], xs[1], "conv"), stride=(s[1], s[2]))
            else:
                y = F.conv2d(xn, self._weight(node.inputs[1], xs[1], "dw"), stride=(s[1], s[2]), groups=c)
            return y.permute(0, 2, 3, 1)
        if op in ("Add", "AddV2", "BiasAdd", "Mul", "Sub"):
            a, b = _t(xs[0], self.tdt), _t(xs[1], self.tdt)
            return {"Add": torch.add, "AddV2": torch.add, "BiasAdd": torch.add, "Mul": torch.mul,
                    "Sub": torch.sub}[op](a, b)
        if op == "Relu":
            return torch.relu(_t(xs[0], self.tdt))
        if op == "Minimum":
            return torch.minimum(_t(xs[0], self.tdt), _t(xs[1], self.tdt))
        if op == "Maximum":
            return torch.maximum(_t(xs[0], self.tdt), _t(xs[1], self.tdt))
        if op == "Mean":
            axes = [int(a) for a in np.asarray(xs[1]).reshape(-1)]
            keep = bool(at["keep_dims"].b) if "keep_dims" in at else False
            return _t(xs[0], self.tdt).mean(dim=axes, keepdim=keep)
        if op == "MatMul":
            return _t(xs[0], self.tdt) @ self._weight(node.inputs[1], xs[1], "mm")
        if op == "Softmax":
            return torch.softmax(_t(xs[0], self.tdt), dim=-1)
        if op == "Sigmoid":
            return torch.sigmoid(_t(xs[0], self.tdt))
        if op in ("Identity", "StopGradient"):
            return xs[0]
        if op == "Reshape":
            return _t(xs[0], self.tdt).reshape([int(d) for d in np.asarray(xs[1]).reshape(-1)])
        raise NotImplementedError("torch lowering has no op %r" % op)


def time_reference_loop(graph_path: str, output_tensor: str, x_nhwc: np.ndarray, threads: int,
                        budget_s: float = 20.0, batch: int = 1):
    """Time the reference's extract loop (facerec_test.py:394 -- one run per image,
    batch 1) on the host: returns (faces_per_s, n_images_timed).  ``batch`` > 1 times the
    best case for the CPU instead."""
    import time
    torch.set_num_threads(threads)
    g = TorchGraphOracle(graph_path)
    inp = "input_1:0"
    g.run(output_tensor, {inp: x_nhwc[:batch]})          # warm-up: weight cache + oneDNN primitives
    g.run(output_tensor, {inp: x_nhwc[:batch]})
    done, t0 = 0, time.perf_counter()
    i = 0
    while True:
        lo = (i * batch) % max(1, x_nhwc.shape[0] - batch + 1)
        g.run(output_tensor, {inp: x_nhwc[lo:lo + batch]})
        done += batch
        i += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return done / el, done


# ----------------------------------------------------------------------------------------------------------------
# The same graph FUSED for the CPU: what a tuned CPU deployment of this network would run --
# per-channel scales folded into the kernels, bias added by the convolution, Relu -> Minimum(6) -> Maximum(0) as ONE
# in-place clamp, channels_last throughout.  Built by walking the chain of the frozen graph (MobileNet is a chain), not
# by importing anything from the product.  Reported NEXT TO the op-by-op figure in bench.py, both kind "port".
# ----------------------------------------------------------------------------------------------------------------
class FusedChainCPU:
    def __init__(self, graph_path: str, output_tensor: str, input_tensor: str = "input_1:0"):
        g = TorchGraphOracle(graph_path)
        out_name, in_name = output_tensor.split(":")[0], input_tensor.split(":")[0]
        chain, cur = [], g.nodes[out_name]
        while cur.name != in_name:
            chain.append(cur)
            cur = g.nodes[cur.inputs[0].split(":")[0].lstrip("^")]
        chain.reverse()

        def const(ref):
            v = g.run(ref, {})
            return np.asarray(v.numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float32)
        self.stages = []
        i = 0
        while i < len(chain):
            nd = chain[i]
            if nd.op in ("Conv2D", "DepthwiseConv2dNative"):
                k = const(nd.inputs[1])
                s = nd.attr["strides"].list_i
                dw = nd.op != "Conv2D"
                w = k.transpose(2, 3, 0, 1) if dw else k.transpose(3, 2, 0, 1)          # -> [O, I/groups, kh, kw]
                w = np.ascontiguousarray(w).astype(np.float32)
                bias = np.zeros(w.shape[0], np.float32)
                same = nd.attr["padding"].s == b"SAME"
                i += 1
                clamp = False
                while i < len(chain) and chain[i].op in ("Mul", "Add", "AddV2", "BiasAdd", "Relu", "Minimum", "Maximum", "Relu6"):
                    f = chain[i]
                    if f.op == "Mul":
                        sc = const(f.inputs[1]).reshape(-1)
                        w *= sc[:, None, None, None]
                        bias *= sc
                    elif f.op in ("Add", "AddV2", "BiasAdd"):
                        bias += const(f.inputs[1]).reshape(-1)
                    elif f.op == "Minimum":
                        assert float(const(f.inputs[1]).reshape(-1)[0]) == 6.0
                        clamp = True
                    elif f.op == "Maximum":
                        assert float(const(f.inputs[1]).reshape(-1)[0]) == 0.0
                    elif f.op == "Relu6":
                        clamp = True
                    i += 1
                wt = torch.from_numpy(w)
                if not dw:
                    wt = wt.contiguous(memory_format=torch.channels_last)
                self.stages.append(("conv", wt, torch.from_numpy(bias), (int(s[1]), int(s[2])), same, w.shape[0] if dw else 1,
                                    (w.shape[2], w.shape[3]), clamp))
                continue
            if nd.op == "Mean":
                axes = [int(a) for a in const(nd.inputs[1]).reshape(-1)]
                self.stages.append(("mean", axes))
            elif nd.op in ("Reshape", "Identity"):
                pass
            else:
                raise NotImplementedError("fused CPU chain: op %r" % nd.op)
            i += 1

    def run(self, x_nhwc: np.ndarray) -> np.ndarray:
        with torch.no_grad():
            x = torch.from_numpy(np.ascontiguousarray(x_nhwc, dtype=np.float32)).permute(0, 3, 1, 2)   # channels_last view
            for st in self.stages:
                if st[0] == "conv":
                    _, w, b, stride, same, groups, (kh, kw), clamp = st
                    if same:
                        _, pt, pb = same_pad(x.shape[2], kh, stride[0])
                        _, pl, pr = same_pad(x.shape[3], kw, stride[1])
                        if pt == pb and pl == pr:
                            x = F.conv2d(x, w, b, stride=stride, padding=(pt, pl), groups=groups)
                        else:
                            x = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, b, stride=stride, groups=groups)
                    else:
                        x = F.conv2d(x, w, b, stride=stride, groups=groups)
                    if clamp:
                        x = x.clamp_(0.0, 6.0)
                else:
                    x = x.mean(dim=[{0: 0, 1: 2, 2: 3, 3: 1}[a] for a in st[1]])          # NHWC axes -> NCHW dims
            return x.reshape(x.shape[0], -1).numpy()


def time_fused_loop(graph_path: str, output_tensor: str, x_nhwc: np.ndarray, threads: int, budget_s: float = 10.0, batch: int = 1):
    """time_reference_loop for the fused chain: (faces_per_s, n_images_timed)."""
    import time
    torch.set_num_threads(threads)
    m = FusedChainCPU(graph_path, output_tensor)
    m.run(x_nhwc[:batch])
    m.run(x_nhwc[:batch])
    done, i, t0 = 0, 0, time.perf_counter()
    while True:
        lo = (i * batch) % max(1, x_nhwc.shape[0] - batch + 1)
        m.run(x_nhwc[lo:lo + batch])
        done += batch
        i += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return done / el, done
