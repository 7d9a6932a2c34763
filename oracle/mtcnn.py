"""Oracle part 6: the MTCNN face-detection cascade as the reference runs it, restated on NumPy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the three nets execute through the
unfused GraphOracle over the reference's own mtcnn.pb, but ``cv2.resize(..., INTER_AREA)`` (OpenCV, version not
pinned by the reference, not installable here) is restated from OpenCV's published algorithm
(modules/imgproc/src/resize.cpp: computeResizeAreaTab / ResizeArea_Invoker / resizeAreaFast_ and the
``area_mode`` branch of the linear path), and nothing in the reference pins detector outputs numerically
(the notebook shows 4 faces on test_image.jpg, AgeGenderIdentityDemo.ipynb:109-125).

Follows facial_analysis.py:334-352 (load_mtcnn), :354-476 (bbreg, generateBoundingBox, nms, pad, rerec) and
:478-604 (mtcnn_detect_faces), including its quirks: images are fed transposed (W,H), boxes live in 1-based
inclusive pixel coordinates, ``np.fix`` truncation, the ``flipud`` branch when exactly one cell fires.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

from .tf_graph import GraphOracle


# ---------------------------------------------------------------------------------------------------
# cv2.resize(src, (dw, dh), interpolation=cv2.INTER_AREA)
# ---------------------------------------------------------------------------------------------------
def _area_tab(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab: list of (dst index, src index, alpha) in OpenCV's order."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, (sx1 - fsx1) / cell))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, 1.0 / cell))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, min(min(fsx2 - sx2, 1.0), cell) / cell))
    return tab


def _round_half_even_u8(a: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(a), 0, 255).astype(np.uint8)     # saturate_cast<uchar>(float) = cvRound + clamp


def cv2_resize_area(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """uint8 or float64 [H,W,C] -> [dh,dw,C] as cv2.resize(..., interpolation=cv2.INTER_AREA) computes it."""
    sh, sw = src.shape[:2]
    if (sh, sw) == (dh, dw):
        return src.copy()
    is_u8 = src.dtype == np.uint8
    wt = np.float32 if is_u8 else np.float64        # area_tab: uchar -> float accumulators, double -> double
    scale_x, scale_y = sw / dw, sh / dh
    if scale_x >= 1 and scale_y >= 1:
        ix, iy = int(round(scale_x)), int(round(scale_y))
        if abs(scale_x - ix) < np.finfo(np.float64).eps and abs(scale_y - iy) < np.finfo(np.float64).eps:
            # resizeAreaFast_: integer box filter
            blk = src.reshape(dh, iy, dw, ix, -1)
            if is_u8:
                s = blk.astype(np.int64).sum(axis=(1, 3))
                if ix == 2 and iy == 2:
                    return ((s + 2) >> 2).astype(np.uint8)              # SIMD 2x2 path
                return _round_half_even_u8(s.astype(np.float32) * np.float32(1.0 / (ix * iy)))
            return blk.sum(axis=(1, 3)) * (1.0 / (ix * iy))
        xtab, ytab = _area_tab(sw, dw, scale_x), _area_tab(sh, dh, scale_y)
        c = src.shape[2]
        out = np.zeros((dh, dw, c), wt)
        S = src.astype(wt)
        prev_dy = -1
        acc = np.zeros((dw, c), wt)
        for dy, sy, beta in ytab:
            buf = np.zeros((dw, c), wt)
            row = S[sy]
            for dx, sx, alpha in xtab:
                buf[dx] += row[sx] * wt(alpha)
            if dy != prev_dy:
                if prev_dy >= 0:
                    out[prev_dy] = acc
                acc = wt(beta) * buf
                prev_dy = dy
            else:
                acc = acc + wt(beta) * buf
        out[prev_dy] = acc
        return _round_half_even_u8(out) if is_u8 else out
    # enlarging in at least one direction: the bilinear path with INTER_AREA's coordinate rule ("area_mode")
    def taps(ssize, dsize, scale):
        inv = 1.0 / scale
        s0 = np.zeros(dsize, np.int64)
        f = np.zeros(dsize, np.float32)
        for d in range(dsize):
            sx = math.floor(d * scale)
            fx = np.float32((d + 1) - (sx + 1) * inv)
            fx = np.float32(0) if fx <= 0 else np.float32(fx - math.floor(fx))
            if sx < 0:
                fx, sx = np.float32(0), 0
            if sx >= ssize - 1:
                fx, sx = np.float32(0), ssize - 1
            s0[d], f[d] = sx, fx
        return s0, np.minimum(s0 + 1, ssize - 1), f
    x0, x1, fx = taps(sw, dw, scale_x)
    y0, y1, fy = taps(sh, dh, scale_y)
    if is_u8:
        ax1 = np.rint(fx * np.float32(2048)).astype(np.int64)
        ax0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
        by1 = np.rint(fy * np.float32(2048)).astype(np.int64)
        by0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
        s = src.astype(np.int64)
        h0 = s[y0][:, x0] * ax0[None, :, None] + s[y0][:, x1] * ax1[None, :, None]
        h1 = s[y1][:, x0] * ax0[None, :, None] + s[y1][:, x1] * ax1[None, :, None]
        v = (((by0[:, None, None] * (h0 >> 4)) >> 16) + ((by1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
        return np.clip(v, 0, 255).astype(np.uint8)
    s = src.astype(np.float64)
    a1, a0 = fx.astype(np.float64), (np.float32(1) - fx).astype(np.float64)     # float coefficients, double data
    b1, b0 = fy.astype(np.float64), (np.float32(1) - fy).astype(np.float64)
    h0 = s[y0][:, x0] * a0[None, :, None] + s[y0][:, x1] * a1[None, :, None]
    h1 = s[y1][:, x0] * a0[None, :, None] + s[y1][:, x1] * a1[None, :, None]
    return h0 * b0[:, None, None] + h1 * b1[:, None, None]


# ---------------------------------------------------------------------------------------------------
# helpers of facial_analysis.py:354-476
# ---------------------------------------------------------------------------------------------------
def bbreg(boundingbox, reg):                                  # :354-367
    if reg.shape[1] == 1:
        reg = np.reshape(reg, (reg.shape[2], reg.shape[3]))
    w = boundingbox[:, 2] - boundingbox[:, 0] + 1
    h = boundingbox[:, 3] - boundingbox[:, 1] + 1
    boundingbox[:, 0:4] = np.stack([boundingbox[:, 0] + reg[:, 0] * w, boundingbox[:, 1] + reg[:, 1] * h,
                                    boundingbox[:, 2] + reg[:, 2] * w, boundingbox[:, 3] + reg[:, 3] * h], axis=1)
    return boundingbox


def generate_bounding_box(imap, reg, scale, t):               # :369-396
    stride, cellsize = 2, 12
    imap = imap.T
    d = [reg[:, :, i].T for i in range(4)]
    y, x = np.where(imap >= t)
    if y.shape[0] == 1:
        d = [np.flipud(v) for v in d]
    score = imap[(y, x)]
    reg = np.stack([v[(y, x)] for v in d], axis=1) if y.size else np.empty((0, 3))
    bb = np.stack([y, x], axis=1)
    q1 = np.fix((stride * bb + 1) / scale)
    q2 = np.fix((stride * bb + cellsize - 1 + 1) / scale)
    return np.hstack([q1, q2, score[:, None], reg]), reg


def nms(boxes, threshold, method, argsort=np.argsort):        # :398-431
    """``argsort``: the reference calls np.argsort(s).  Its order among EQUAL scores is whatever the installed NumPy does:
    the NumPy of the reference's time (<= 1.24) finishes lists of <= 16 elements with a stable insertion sort (ascending index
    among ties), NumPy >= 1.25 on AVX-512/AVX2 hosts uses a SIMD sort with another tie order.  Tests that pin the tie rule
    pass ``lambda s: np.argsort(s, kind="stable")`` = the old behaviour on the list sizes that occur."""
    if boxes.size == 0:
        return np.empty((0, 3))
    x1, y1, x2, y2, s = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3], boxes[:, 4]
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = argsort(s)
    pick = []
    while order.size > 0:
        i = order[-1]
        pick.append(i)
        idx = order[:-1]
        w = np.maximum(0.0, np.minimum(x2[i], x2[idx]) - np.maximum(x1[i], x1[idx]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[idx]) - np.maximum(y1[i], y1[idx]) + 1)
        inter = w * h
        o = inter / np.minimum(area[i], area[idx]) if method == 'Min' else inter / (area[i] + area[idx] - inter)
        order = order[np.where(o <= threshold)]
    return np.asarray(pick, dtype=np.int16)


def pad(total_boxes, w, h):                                   # :433-465
    tmpw = (total_boxes[:, 2] - total_boxes[:, 0] + 1).astype(np.int32)
    tmph = (total_boxes[:, 3] - total_boxes[:, 1] + 1).astype(np.int32)
    n = total_boxes.shape[0]
    dx, dy = np.ones(n, np.int32), np.ones(n, np.int32)
    edx, edy = tmpw.copy(), tmph.copy()
    x, y = total_boxes[:, 0].astype(np.int32), total_boxes[:, 1].astype(np.int32)
    ex, ey = total_boxes[:, 2].astype(np.int32), total_boxes[:, 3].astype(np.int32)
    m = ex > w
    edx[m] = -ex[m] + w + tmpw[m]
    ex[m] = w
    m = ey > h
    edy[m] = -ey[m] + h + tmph[m]
    ey[m] = h
    m = x < 1
    dx[m] = 2 - x[m]
    x[m] = 1
    m = y < 1
    dy[m] = 2 - y[m]
    y[m] = 1
    return dy, edy, dx, edx, y, ey, x, ex, tmpw, tmph


def rerec(bbox):                                              # :467-476
    h = bbox[:, 3] - bbox[:, 1]
    w = bbox[:, 2] - bbox[:, 0]
    side = np.maximum(w, h)
    bbox[:, 0] = bbox[:, 0] + w * 0.5 - side * 0.5
    bbox[:, 1] = bbox[:, 1] + h * 0.5 - side * 0.5
    bbox[:, 2:4] = bbox[:, 0:2] + side[:, None]
    return bbox


class OracleMTCNN:
    def __init__(self, mtcnn_pb: str, minsize: int = 32, compute_dtype=np.float32):
        self.g = GraphOracle(mtcnn_pb, compute_dtype)
        self.minsize = minsize
        self.dt = compute_dtype

    # :334-352
    def pnet(self, img):
        r = self.g.run(['pnet/conv4-2/BiasAdd:0', 'pnet/prob1:0'], {'pnet/input:0': img})
        return [np.asarray(a, np.float32) for a in r]

    def rnet(self, img):
        r = self.g.run(['rnet/conv5-2/conv5-2:0', 'rnet/prob1:0'], {'rnet/input:0': img})
        return [np.asarray(a, np.float32) for a in r]

    def onet(self, img):
        r = self.g.run(['onet/conv6-2/conv6-2:0', 'onet/conv6-3/conv6-3:0', 'onet/prob1:0'], {'onet/input:0': img})
        return [np.asarray(a, np.float32) for a in r]

    def scales(self, h, w):                                   # :489-499
        factor, m = 0.709, 12.0 / self.minsize
        minl = min(h, w) * m
        out, k = [], 0
        while minl >= 12:
            out.append(m * np.power(factor, k))
            minl *= factor
            k += 1
        return out

    def _crops(self, img, boxes, size):                       # :540-548 / :570-578
        h, w = img.shape[:2]
        dy, edy, dx, edx, y, ey, x, ex, tmpw, tmph = pad(boxes.copy(), w, h)
        n = boxes.shape[0]
        temp = np.zeros((size, size, 3, n))
        for k in range(n):
            tmp = np.zeros((int(tmph[k]), int(tmpw[k]), 3))
            tmp[dy[k] - 1:edy[k], dx[k] - 1:edx[k], :] = img[y[k] - 1:ey[k], x[k] - 1:ex[k], :]
            temp[:, :, :, k] = cv2_resize_area(tmp, size, size)
        temp = (temp - 127.5) * 0.0078125
        return np.transpose(temp, (3, 1, 0, 2))

    def detect(self, img) -> Tuple[np.ndarray, np.ndarray]:   # :478-604
        threshold = [0.6, 0.7, 0.9]
        total_boxes = np.empty((0, 9))
        points = np.array([])
        h, w = img.shape[0], img.shape[1]
        for scale in self.scales(h, w):
            hs, ws = int(np.ceil(h * scale)), int(np.ceil(w * scale))
            im_data = (cv2_resize_area(img, ws, hs) - 127.5) * 0.0078125
            out = self.pnet(np.transpose(im_data[None], (0, 2, 1, 3)))
            out0 = np.transpose(out[0], (0, 2, 1, 3))
            out1 = np.transpose(out[1], (0, 2, 1, 3))
            boxes, _ = generate_bounding_box(out1[0, :, :, 1].copy(), out0[0].copy(), scale, threshold[0])
            pick = nms(boxes.copy(), 0.5, 'Union')
            if boxes.size > 0 and pick.size > 0:
                total_boxes = np.append(total_boxes, boxes[pick, :], axis=0)
        if total_boxes.shape[0] > 0:
            pick = nms(total_boxes.copy(), 0.7, 'Union')
            total_boxes = total_boxes[pick, :]
            regw = total_boxes[:, 2] - total_boxes[:, 0]
            regh = total_boxes[:, 3] - total_boxes[:, 1]
            total_boxes = np.stack([total_boxes[:, 0] + total_boxes[:, 5] * regw, total_boxes[:, 1] + total_boxes[:, 6] * regh,
                                    total_boxes[:, 2] + total_boxes[:, 7] * regw, total_boxes[:, 3] + total_boxes[:, 8] * regh,
                                    total_boxes[:, 4]], axis=1)
            total_boxes = rerec(total_boxes.copy())
            total_boxes[:, 0:4] = np.fix(total_boxes[:, 0:4]).astype(np.int32)
        if total_boxes.shape[0] > 0:
            out = self.rnet(self._crops(img, total_boxes, 24))
            out0, out1 = np.transpose(out[0]), np.transpose(out[1])
            score = out1[1, :]
            ipass = np.where(score > threshold[1])
            total_boxes = np.hstack([total_boxes[ipass[0], 0:4].copy(), np.expand_dims(score[ipass].copy(), 1)])
            mv = out0[:, ipass[0]]
            if total_boxes.shape[0] > 0:
                pick = nms(total_boxes, 0.7, 'Union')
                total_boxes = total_boxes[pick, :]
                total_boxes = bbreg(total_boxes.copy(), np.transpose(mv[:, pick]))
                total_boxes = rerec(total_boxes.copy())
        if total_boxes.shape[0] > 0:
            total_boxes = np.fix(total_boxes).astype(np.int32)
            out = self.onet(self._crops(img, total_boxes, 48))
            out0, out1, out2 = np.transpose(out[0]), np.transpose(out[1]), np.transpose(out[2])
            score = out2[1, :]
            ipass = np.where(score > threshold[2])
            points = out1[:, ipass[0]]
            total_boxes = np.hstack([total_boxes[ipass[0], 0:4].copy(), np.expand_dims(score[ipass].copy(), 1)])
            mv = out0[:, ipass[0]]
            bw = total_boxes[:, 2] - total_boxes[:, 0] + 1
            bh = total_boxes[:, 3] - total_boxes[:, 1] + 1
            points[0:5, :] = np.tile(bw, (5, 1)) * points[0:5, :] + np.tile(total_boxes[:, 0], (5, 1)) - 1
            points[5:10, :] = np.tile(bh, (5, 1)) * points[5:10, :] + np.tile(total_boxes[:, 1], (5, 1)) - 1
            if total_boxes.shape[0] > 0:
                total_boxes = bbreg(total_boxes.copy(), np.transpose(mv))
                pick = nms(total_boxes.copy(), 0.7, 'Min')
                total_boxes = total_boxes[pick, :]
                points = points[:, pick]
        return total_boxes, points
