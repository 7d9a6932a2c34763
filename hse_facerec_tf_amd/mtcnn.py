"""MTCNN face detection on libhsefr: the step before the age/gender path.

Replaces ``FacialImageProcessing.load_mtcnn`` / ``mtcnn_detect_faces`` (facial_analysis.py:334-352, 478-604).
The three nets of the reference's ``mtcnn.pb`` (P-Net fully convolutional over an image pyramid, R-Net on 24x24
crops, O-Net on 48x48 crops) are read with the TF-free GraphDef reader and executed by a small device-side graph
walker that fuses the patterns the file is made of:

    BiasAdd(Conv2D | MatMul) [+ Relu(v) + alpha * -Relu(-v)]   -> hsefr_conv2d_direct (bias + PReLU fused; a MatMul is
                                                                 the VALID conv spanning the flattened map)
    MaxPool                                                     -> hsefr_maxpool_f32
    RealDiv(Exp(x - Max(x)), Sum(Exp(..)))                      -> hsefr_softmax over the last axis

The image pyramid and the R-Net / O-Net crops (cv2.resize INTER_AREA, facial_analysis.py:507,546,575) are cut on the device
from ONE upload of the frame (csrc/area_resize.hip: hsefr_mtcnn_pyramid_level / hsefr_mtcnn_crops, normalisation and the
(W,H) transposition fused in); every P-Net level is launched before the first result is read back.  The box bookkeeping
(threshold, NMS over a few hundred boxes, regression, squaring, window clipping) is host NumPy, written to give the
reference's numbers including its conventions: boxes are 1-based inclusive, ``np.fix`` truncation.
``device_resize=False`` keeps the round-1 path (host resizes through preprocess.resize_area).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _lib, ops, preprocess
from .graphdef import Graph, GraphNode, read_graph

MTCNN_PB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models", "mtcnn.pb")


class _DeviceNet:
    """Evaluate tensors of a small frozen CNN on the GPU, fusing bias / PReLU / softmax sub-graphs."""

    def __init__(self, graph: Graph, device=None):
        torch = _lib.require_gpu()
        self.g = graph
        self._torch = torch
        self.device = _lib.cuda_device(device)
        self._const: Dict[str, object] = {}
        self._programs: Dict[tuple, tuple] = {}

    def _dev(self, node: GraphNode, reshape=None):
        key = node.name + ("" if reshape is None else str(reshape))
        t = self._const.get(key)
        if t is None:
            a = np.ascontiguousarray(self._const_np(node), dtype=np.float32)
            if reshape is not None:
                a = a.reshape(reshape)
            t = self._torch.from_numpy(a).to(self.device)
            self._const[key] = t
        return t

    def _const_np(self, node: GraphNode) -> np.ndarray:
        while node.op == "Identity":
            node = self.g.data_inputs(node)[0][0]
        if node.op != "Const":
            raise NotImplementedError("expected a constant at %s" % node.name)
        return self.g.const_value(node)

    def _is_const(self, node: GraphNode) -> bool:
        while node.op == "Identity":
            node = self.g.data_inputs(node)[0][0]
        return node.op == "Const"

    def _ins(self, node: GraphNode) -> List[GraphNode]:
        return [n for n, _ in self.g.data_inputs(node)]

    # -- pattern matchers ---------------------------------------------------------------------------------
    def _match_prelu(self, node: GraphNode):
        """Add(Relu(v), Mul(alpha, Neg(Relu(Neg(v))))) -> (v, alpha) or None."""
        if node.op not in ("Add", "AddV2"):
            return None
        a, b = self._ins(node)
        for pos, neg in ((a, b), (b, a)):
            if pos.op != "Relu" or neg.op != "Mul":
                continue
            v = self._ins(pos)[0]
            m0, m1 = self._ins(neg)
            for alpha, chain in ((m0, m1), (m1, m0)):
                if not self._is_const(alpha) or chain.op != "Neg":
                    continue
                r = self._ins(chain)[0]
                if r.op != "Relu":
                    continue
                ng = self._ins(r)[0]
                if ng.op == "Neg" and self._ins(ng)[0].name == v.name:
                    return v, alpha
        return None

    def _match_softmax(self, node: GraphNode):
        """RealDiv(Exp(Sub(x, Max(x))), Sum(Exp(Sub(x, Max(x))))) -> x or None."""
        if node.op != "RealDiv":
            return None
        e, s = self._ins(node)
        if e.op != "Exp" or s.op != "Sum" or self._ins(s)[0].name != e.name:
            return None
        sub = self._ins(e)[0]
        if sub.op != "Sub":
            return None
        x, mx = self._ins(sub)
        if mx.op != "Max" or self._ins(mx)[0].name != x.name:
            return None
        return x

    # -- evaluation ------------------------------------------------------------------------------------------
    # The graph is walked ONCE per (fetches, input) pair: pattern matching, attribute parsing and constant uploads produce a linear
    # program of (output name, callable, input names); every later run just executes it (round 3: the recursive interpreter of
    # rounds 1-2 re-matched the PReLU / softmax sub-graphs on every call -- 0.5 ms of host time per frame for ~60 launches).
    def run(self, fetches: List[str], input_name: str, x):
        key = (tuple(fetches), input_name)
        prog = self._programs.get(key)
        in_name = self.g.get_tensor_by_name(input_name)[0].name
        if prog is None:
            steps: List[Tuple[str, object, Tuple[str, ...]]] = []
            done = {in_name}
            outs = []
            for f in fetches:
                node = self.g.get_tensor_by_name(f)[0]
                self._compile(node, steps, done)
                outs.append(node.name)
            prog = self._programs[key] = (steps, outs)
        steps, outs = prog
        vals: Dict[str, object] = {in_name: x}
        for name, fn, ins in steps:
            vals[name] = fn(*[vals[i] for i in ins])
        return [vals[o] for o in outs]

    def _compile(self, node: GraphNode, steps, done) -> None:
        if node.name in done:
            return
        fn, srcs = self._compile_node(node)
        for sn in srcs:
            self._compile(sn, steps, done)
        steps.append((node.name, fn, tuple(sn.name for sn in srcs)))
        done.add(node.name)

    def _compile_linear(self, node: GraphNode, alpha_node: Optional[GraphNode]):
        """node = BiasAdd(Conv2D | MatMul) (or a bare Conv2D/MatMul): one fused kernel launch."""
        bias = None
        if node.op == "BiasAdd":
            lin, bnode = self._ins(node)
            bias = self._dev(bnode)
        else:
            lin = node
        alpha = None if alpha_node is None else self._dev(alpha_node)
        src, wnode = self._ins(lin)
        if lin.op == "Conv2D":
            stride, padding, wdev = lin.attr_ints("strides")[1], lin.attr_s("padding"), self._dev(wnode)
            return (lambda xin: ops.conv2d_direct(xin, wdev, bias, alpha, stride, padding)), [src]
        if lin.op == "MatMul":
            w = self._const_np(wnode)
            cout = int(w.shape[1])
            wdev = self._dev(wnode, (1, 1, w.shape[0], w.shape[1]))

            def matmul(xin):
                n = xin.shape[0]
                flat = xin.reshape(n, 1, 1, -1).contiguous()
                return ops.conv2d_direct(flat, wdev, bias, alpha, 1, "VALID").reshape(n, cout)
            return matmul, [src]
        raise NotImplementedError("%s: op %s" % (lin.name, lin.op))

    def _compile_node(self, node: GraphNode):
        """-> (callable(*input tensors), [input nodes])."""
        op = node.op
        pr = self._match_prelu(node)
        if pr is not None:
            return self._compile_linear(pr[0], pr[1])
        sm = self._match_softmax(node)
        if sm is not None:
            return (lambda x: ops.softmax(x.reshape(-1, x.shape[-1]).contiguous()).reshape(x.shape)), [sm]
        if op in ("BiasAdd", "Conv2D", "MatMul"):
            return self._compile_linear(node, None)
        if op == "MaxPool":
            k, st, padding = node.attr_ints("ksize")[1], node.attr_ints("strides")[1], node.attr_s("padding")
            return (lambda x: ops.maxpool(x, k, st, padding)), [self._ins(node)[0]]
        if op == "Reshape":
            shape = [int(d) for d in self._const_np(self._ins(node)[1]).reshape(-1)]
            return (lambda x: x.reshape(shape).contiguous()), [self._ins(node)[0]]
        if op == "Identity":
            return (lambda x: x), [self._ins(node)[0]]
        raise NotImplementedError("MTCNN graph walker: no kernel for op %s (%s)" % (op, node.name))


# ---- box utilities (facial_analysis.py:354-476 semantics) -------------------------------------------------------
def _iou_suppress(boxes: np.ndarray, threshold: float, use_min: bool) -> np.ndarray:
    """Greedy NMS by descending score; overlap = IoU, or intersection / smaller area ('Min')."""
    if boxes.size == 0:
        return np.empty((0,), np.int64)
    x1, y1, x2, y2, score = (boxes[:, i] for i in range(5))
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    # the reference: `I = np.argsort(s)`, pick I[-1] (facial_analysis.py:398-403).  Among bit-equal scores (softmax saturated
    # to 1.0f on clear faces) NumPy's default sort leaves lists of <= 16 elements -- the usual R-/O-Net list -- in ascending
    # index order, so the HIGHEST index is picked first.  A stable sort makes that the rule for every list size; the device
    # kernel (csrc/mtcnn_post.hip make_key) uses the same rule.
    order = np.argsort(score.astype(np.float32), kind="stable")
    keep = []
    while order.size:
        top, rest = order[-1], order[:-1]
        keep.append(top)
        iw = np.maximum(0.0, np.minimum(x2[top], x2[rest]) - np.maximum(x1[top], x1[rest]) + 1)
        ih = np.maximum(0.0, np.minimum(y2[top], y2[rest]) - np.maximum(y1[top], y1[rest]) + 1)
        inter = iw * ih
        ov = inter / (np.minimum(area[top], area[rest]) if use_min else (area[top] + area[rest] - inter))
        order = rest[ov <= threshold]
    return np.asarray(keep, np.int64)


def _regress(boxes: np.ndarray, reg: np.ndarray) -> np.ndarray:
    out = boxes.copy()
    w = boxes[:, 2] - boxes[:, 0] + 1
    h = boxes[:, 3] - boxes[:, 1] + 1
    out[:, 0] = boxes[:, 0] + reg[:, 0] * w
    out[:, 1] = boxes[:, 1] + reg[:, 1] * h
    out[:, 2] = boxes[:, 2] + reg[:, 2] * w
    out[:, 3] = boxes[:, 3] + reg[:, 3] * h
    return out


def _square(boxes: np.ndarray) -> np.ndarray:
    out = boxes.copy()
    w = boxes[:, 2] - boxes[:, 0]
    h = boxes[:, 3] - boxes[:, 1]
    side = np.maximum(w, h)
    out[:, 0] = boxes[:, 0] + w * 0.5 - side * 0.5
    out[:, 1] = boxes[:, 1] + h * 0.5 - side * 0.5
    out[:, 2] = out[:, 0] + side
    out[:, 3] = out[:, 1] + side
    return out


def _crop_windows(boxes: np.ndarray, img_w: int, img_h: int):
    """1-based inclusive source window of each box clipped to the frame, and where it lands in the box-sized tile."""
    bw = (boxes[:, 2] - boxes[:, 0] + 1).astype(np.int32)
    bh = (boxes[:, 3] - boxes[:, 1] + 1).astype(np.int32)
    x1, y1 = boxes[:, 0].astype(np.int32), boxes[:, 1].astype(np.int32)
    x2, y2 = boxes[:, 2].astype(np.int32), boxes[:, 3].astype(np.int32)
    tx1, ty1 = np.ones_like(x1), np.ones_like(y1)
    tx2, ty2 = bw.copy(), bh.copy()
    over = x2 > img_w
    tx2[over] = img_w - x2[over] + bw[over]
    x2 = np.where(over, img_w, x2)
    over = y2 > img_h
    ty2[over] = img_h - y2[over] + bh[over]
    y2 = np.where(over, img_h, y2)
    under = x1 < 1
    tx1[under] = 2 - x1[under]
    x1 = np.where(under, 1, x1)
    under = y1 < 1
    ty1[under] = 2 - y1[under]
    y1 = np.where(under, 1, y1)
    return bw, bh, x1, y1, x2, y2, tx1, ty1, tx2, ty2


class MTCNNDetector:
    """callable(img_rgb_uint8) -> (bounding_boxes [n,5] = x1,y1,x2,y2,score; points [10,n])."""

    THRESHOLDS = (0.6, 0.7, 0.9)       # facial_analysis.py:481
    FACTOR = 0.709                     # :483

    def __init__(self, mtcnn_pb: Optional[str] = None, minsize: int = 32, device=None, device_resize: bool = True,
                 device_boxes: Optional[bool] = None):
        """device_resize: pyramid levels and crops resampled on the GPU (csrc/area_resize.hip); device_boxes (default: as
        device_resize): candidate generation, NMS, box regression, squaring, crop windows and landmarks on the GPU too
        (csrc/mtcnn_post.hip) -- the host then only reads three box counts per frame."""
        torch = _lib.require_gpu()
        self._torch = torch
        self.minsize = minsize
        self.device_resize = device_resize
        self.device_boxes = device_resize if device_boxes is None else bool(device_boxes)
        if self.device_boxes and not device_resize:
            raise ValueError("device_boxes needs device_resize (the crops are cut from the frame on the device)")
        self.host_fallbacks = 0            # frames redone on the host because a candidate list overflowed
        self._work = None                  # cap-sized work tensors of the device box logic, allocated once (_detect_device)
        self.device = _lib.cuda_device(device)
        self.net = _DeviceNet(read_graph(mtcnn_pb or MTCNN_PB), self.device)

    # the three sess.run lambdas of load_mtcnn (:347-349)
    def pnet(self, img):
        return self.net.run(['pnet/conv4-2/BiasAdd:0', 'pnet/prob1:0'], 'pnet/input:0', img)

    def rnet(self, img):
        return self.net.run(['rnet/conv5-2/conv5-2:0', 'rnet/prob1:0'], 'rnet/input:0', img)

    def onet(self, img):
        return self.net.run(['onet/conv6-2/conv6-2:0', 'onet/conv6-3/conv6-3:0', 'onet/prob1:0'], 'onet/input:0', img)

    def pyramid_scales(self, h: int, w: int) -> List[float]:
        m = 12.0 / self.minsize
        side = min(h, w) * m
        scales, k = [], 0
        while side >= 12:
            scales.append(m * np.power(self.FACTOR, k))
            side *= self.FACTOR
            k += 1
        return scales

    def _to_device(self, a: np.ndarray):
        return self._torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def _level_device(self, frame, h: int, w: int, hs: int, ws: int):
        """One pyramid level of the uploaded frame (``frame``: CUDA uint8 [h, w, 3] -- an argument, not detector state, so one
        detector can serve several threads), normalised and transposed: CUDA float32 [1, ws, hs, 3]."""
        torch = self._torch
        out = torch.empty((1, ws, hs, 3), dtype=torch.float32, device=self.device)
        with _lib.on_device(out):
            _lib.check(_lib.lib().hsefr_mtcnn_pyramid_level(frame.data_ptr(), out.data_ptr(), h, w, hs, ws,
                                                            _lib.current_stream_ptr()), "hsefr_mtcnn_pyramid_level")
        return out

    def _stage1(self, img: np.ndarray, frame=None) -> np.ndarray:
        h, w = img.shape[:2]
        found = [np.empty((0, 9))]
        scales = self.pyramid_scales(h, w)
        maps = []
        for scale in scales:                                         # launch every level first ...
            hs, ws = int(np.ceil(h * scale)), int(np.ceil(w * scale))
            if self.device_resize:
                x = self._level_device(frame, h, w, hs, ws)
            else:
                level = (preprocess.resize_area(img, ws, hs) - 127.5) * 0.0078125
                x = self._to_device(np.transpose(level, (1, 0, 2))[None])           # nets see (W, H)
            maps.append(self.pnet(x))
        for scale, (reg_t, prob_t) in zip(scales, maps):             # ... then read the maps back
            prob = prob_t[0, :, :, 1].cpu().numpy()                  # [W', H']
            reg = reg_t[0].cpu().numpy()                             # [W', H', 4]
            xi, yi = np.nonzero(prob >= self.THRESHOLDS[0])
            if xi.size == 0:
                continue
            score = prob[xi, yi]
            # the reference flips the regression maps when exactly one cell fires (:383-387)
            rr = reg[prob.shape[0] - 1 - xi, yi] if xi.size == 1 else reg[xi, yi]
            cell = np.stack([xi, yi], axis=1)
            boxes = np.hstack([np.fix((2 * cell + 1) / scale), np.fix((2 * cell + 12) / scale), score[:, None], rr])
            keep = _iou_suppress(boxes, 0.5, False)
            if keep.size:
                found.append(boxes[keep])
        return np.concatenate(found, axis=0)

    def _crops(self, img: np.ndarray, boxes: np.ndarray, size: int, frame=None):
        h, w = img.shape[:2]
        bw, bh, x1, y1, x2, y2, tx1, ty1, tx2, ty2 = _crop_windows(boxes, w, h)
        if self.device_resize:
            torch = self._torch
            n = boxes.shape[0]
            tab = np.stack([x1, y1, x2, y2, tx1, ty1, bw, bh], axis=1).astype(np.int32)
            d_tab = torch.from_numpy(np.ascontiguousarray(tab)).to(self.device)
            out = torch.empty((n, size, size, 3), dtype=torch.float32, device=self.device)
            with _lib.on_device(out):
                _lib.check(_lib.lib().hsefr_mtcnn_crops(frame.data_ptr(), d_tab.data_ptr(), out.data_ptr(), h, w, n, size,
                                                        _lib.current_stream_ptr()), "hsefr_mtcnn_crops")
            return out
        out = np.zeros((boxes.shape[0], size, size, 3))
        for k in range(boxes.shape[0]):
            tile = np.zeros((int(bh[k]), int(bw[k]), 3))
            tile[ty1[k] - 1:ty2[k], tx1[k] - 1:tx2[k], :] = img[y1[k] - 1:y2[k], x1[k] - 1:x2[k], :]
            out[k] = preprocess.resize_area(tile, size, size)
        out = (out - 127.5) * 0.0078125
        return self._to_device(np.transpose(out, (0, 2, 1, 3)))      # [n, W, H, 3]

    # ---- the cascade with its box logic on the device (csrc/mtcnn_post.hip) ------------------------------------------------
    def _work_tensors(self, cap: int):
        """The cap-sized lists of the device box logic, allocated once per detector and thread (work on one stream is ordered,
        so a frame may overwrite what the previous frame of the same thread left)."""
        import threading
        torch, dev = self._torch, self.device
        if self._work is None:
            self._work = threading.local()
        wk = self._work
        if getattr(wk, "cap", None) != cap:
            wk.cap = cap
            wk.found = torch.empty((cap, 9), dtype=torch.float64, device=dev)
            wk.boxes = [torch.empty((cap, 5), dtype=torch.float64, device=dev) for _ in range(3)]
            wk.tab = torch.empty((cap, 8), dtype=torch.int32, device=dev)
            wk.points = torch.empty((cap, 10), dtype=torch.float32, device=dev)
        return wk

    def _detect_device(self, img: np.ndarray, frame):
        """Returns (boxes, points), or None when a candidate list overflowed the device capacity (caller falls back)."""
        torch, L = self._torch, _lib.lib()
        h, w = img.shape[:2]
        cap = int(L.hsefr_mtcnn_post_capacity())
        dev = self.device
        with _lib.on_device(frame):
            st = _lib.current_stream_ptr()
            wk = self._work_tensors(cap)
            counters = torch.zeros(8, dtype=torch.int32, device=dev)
            found, (boxes1, boxes2, boxes3), tab, points3 = wk.found, wk.boxes, wk.tab, wk.points
            thr = [float(np.float32(t)) for t in self.THRESHOLDS]
            for scale in self.pyramid_scales(h, w):
                hs, ws = int(np.ceil(h * scale)), int(np.ceil(w * scale))
                reg_t, prob_t = self.pnet(self._level_device(frame, h, w, hs, ws))
                reg_t, prob_t = reg_t.contiguous(), prob_t.contiguous()
                _lib.check(L.hsefr_mtcnn_stage1_level(prob_t.data_ptr(), reg_t.data_ptr(), int(prob_t.shape[1]), int(prob_t.shape[2]),
                                                      float(scale), thr[0], found.data_ptr(), counters.data_ptr(), st), "hsefr_mtcnn_stage1_level")
            _lib.check(L.hsefr_mtcnn_stage1_finish(found.data_ptr(), counters.data_ptr(), boxes1.data_ptr(), tab.data_ptr(), w, h, st),
                       "hsefr_mtcnn_stage1_finish")
            c = counters.cpu().numpy()
            if c[4]:
                return None
            n1 = int(c[1])
            if n1 == 0:
                return np.empty((0, 9)), np.array([])
            crops = torch.empty((n1, 24, 24, 3), dtype=torch.float32, device=dev)
            _lib.check(L.hsefr_mtcnn_crops(frame.data_ptr(), tab.data_ptr(), crops.data_ptr(), h, w, n1, 24, st), "hsefr_mtcnn_crops")
            reg_t, prob_t = self.rnet(crops)
            reg_t, prob_t = reg_t.contiguous(), prob_t.contiguous()
            _lib.check(L.hsefr_mtcnn_stage_finish(2, boxes1.data_ptr(), n1, prob_t.data_ptr(), reg_t.data_ptr(), None, thr[1], boxes2.data_ptr(),
                                                  tab.data_ptr(), None, counters.data_ptr(), w, h, st), "hsefr_mtcnn_stage_finish")
            n2 = int(counters.cpu().numpy()[2])
            if n2 == 0:
                return np.empty((0, 5)), np.array([])
            crops = torch.empty((n2, 48, 48, 3), dtype=torch.float32, device=dev)
            _lib.check(L.hsefr_mtcnn_crops(frame.data_ptr(), tab.data_ptr(), crops.data_ptr(), h, w, n2, 48, st), "hsefr_mtcnn_crops")
            reg_t, pts_t, prob_t = self.onet(crops)
            reg_t, pts_t, prob_t = reg_t.contiguous(), pts_t.contiguous(), prob_t.contiguous()
            _lib.check(L.hsefr_mtcnn_stage_finish(3, boxes2.data_ptr(), n2, prob_t.data_ptr(), reg_t.data_ptr(), pts_t.data_ptr(), thr[2],
                                                  boxes3.data_ptr(), None, points3.data_ptr(), counters.data_ptr(), w, h, st), "hsefr_mtcnn_stage_finish")
            n3 = int(counters.cpu().numpy()[3])
            if n3 == 0:
                return np.empty((0, 5)), np.empty((10, 0), np.float32)
            return boxes3[:n3].cpu().numpy(), np.ascontiguousarray(points3[:n3].cpu().numpy().T)

    def __call__(self, img: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        img = np.asarray(img)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("the detector takes an RGB frame [H, W, 3]")
        frame = None
        if self.device_resize:
            if img.dtype != np.uint8:      # the device resampler is OpenCV's 8-bit path; float frames: MTCNNDetector(device_resize=False)
                raise ValueError("device_resize takes a uint8 RGB frame [H, W, 3]; construct the detector with device_resize=False "
                                 "for %s frames" % img.dtype)
            frame = self._torch.from_numpy(np.ascontiguousarray(img)).to(self.device)     # local: no per-frame detector state
        if self.device_boxes:
            res = self._detect_device(img, frame)
            if res is not None:
                return res
            self.host_fallbacks += 1
        points = np.array([])
        boxes = self._stage1(img, frame)
        if boxes.shape[0]:
            boxes = boxes[_iou_suppress(boxes, 0.7, False)]
            rw, rh = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
            boxes = np.stack([boxes[:, 0] + boxes[:, 5] * rw, boxes[:, 1] + boxes[:, 6] * rh, boxes[:, 2] + boxes[:, 7] * rw,
                              boxes[:, 3] + boxes[:, 8] * rh, boxes[:, 4]], axis=1)
            boxes = _square(boxes)
            boxes[:, 0:4] = np.fix(boxes[:, 0:4]).astype(np.int32)
        if boxes.shape[0]:
            reg_t, prob_t = self.rnet(self._crops(img, boxes, 24, frame))
            score = prob_t[:, 1].cpu().numpy()
            reg = reg_t.cpu().numpy()
            ok = np.nonzero(score > self.THRESHOLDS[1])[0]
            boxes = np.hstack([boxes[ok, 0:4], score[ok][:, None]])
            reg = reg[ok]
            if boxes.shape[0]:
                keep = _iou_suppress(boxes, 0.7, False)
                boxes = _square(_regress(boxes[keep], reg[keep]))
        if boxes.shape[0]:
            boxes = np.fix(boxes).astype(np.int32)
            reg_t, pts_t, prob_t = self.onet(self._crops(img, boxes, 48, frame))
            score = prob_t[:, 1].cpu().numpy()
            ok = np.nonzero(score > self.THRESHOLDS[2])[0]
            points = pts_t.cpu().numpy()[ok].T                       # [10, n]
            reg = reg_t.cpu().numpy()[ok]
            boxes = np.hstack([boxes[ok, 0:4], score[ok][:, None]])
            bw = boxes[:, 2] - boxes[:, 0] + 1
            bh = boxes[:, 3] - boxes[:, 1] + 1
            points[0:5, :] = bw[None, :] * points[0:5, :] + boxes[:, 0][None, :] - 1
            points[5:10, :] = bh[None, :] * points[5:10, :] + boxes[:, 1][None, :] - 1
            if boxes.shape[0]:
                boxes = _regress(boxes, reg)
                keep = _iou_suppress(boxes, 0.7, True)
                boxes, points = boxes[keep], points[:, keep]
        return boxes, points
