"""Drop-in for the age/gender/identity part of the reference's ``FacialImageProcessing``
(age_gender_identity/facial_analysis.py:36-130,225-294): ``age_gender_fun(img)``,
``process_image(draw)``, ``close()``, ``is_male()`` -- on libhsefr instead of ``tf.Session``.

Face *detection* is the step before the hot path: with ``mtcnn_detector=True`` the MTCNN
cascade of the reference's mtcnn.pb runs on the GPU (hse_facerec_tf_amd/mtcnn.py; facial_analysis.py:334-604);
any other detector (e.g. the OpenCV LBP cascade of :217-222) can be injected as a callable
``detector(img_rgb) -> (bounding_boxes, points)``.

Added for throughput: ``age_gender_batch`` -- all faces of a frame/album in ONE forward (the
reference runs one ``sess.run`` per face, facial_analysis.py:266-271).
"""
from __future__ import annotations

import os
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, preprocess
from .engine import Engine
from .graphdef import read_graph
from .lowering import OUT_AGE, OUT_FEATURES, OUT_GENDER, lower_graph
from .tf_inference import AGE_GENDER_PB


def decode_age(age_preds: np.ndarray, min_age: int = 1):
    """facial_analysis.py:112-124: renormalised expectation over the two most probable age bins."""
    indices = age_preds.argsort()[::-1][:2]
    norm_preds = age_preds[indices] / np.sum(age_preds[indices])
    res_age = min_age
    for age, probab in zip(indices, norm_preds):
        res_age += age * probab
    return res_age, indices, norm_preds


class FacialImageProcessing:
    # minsize: minimum of faces' size
    def __init__(self, print_stat=False, mtcnn_detector=True, minsize=32, model_file: Optional[str] = None,
                 detector: Optional[Callable] = None, max_batch: int = 64, device: Optional[int] = None,
                 device_preprocess: bool = True, latency_plan: bool = True):
        # latency_plan: age_gender_fun / process_image -- the reference's one-face-per-run calls (:93-129, :225-294) -- run the
        # small-batch lowering of the graph (Engine.__init__: one face 0.17 instead of 0.26 ms on the device); age_gender_batch,
        # the bulk entry, keeps the one plan for every batch size unless asked (latency=True).  Results of the two plans differ in
        # the last bits (3e-7 relative; both are tested against the oracle at the same tolerance).
        self.latency_plan = bool(latency_plan)
        self.device_preprocess = device_preprocess
        self.mtcnn_detector = mtcnn_detector
        self.print_stat = print_stat
        self.minsize = minsize
        self.detector = detector
        if detector is None and mtcnn_detector:        # facial_analysis.py:59-61: the MTCNN cascade of mtcnn.pb
            from .mtcnn import MTCNNDetector
            self.detector = MTCNNDetector(minsize=minsize, device=device)
        # facial_analysis.py:45 loads a sibling .pb; the one that ships with the reference checkout is
        # the quantised age_gender_tf2_new-01-0.14-0.92 model.
        self.model_file = model_file or AGE_GENDER_PB
        self.age_gender_fun = self.load_age_gender(self.model_file, max_batch, device)

    def close(self):                          # facial_analysis.py:73-74
        self.sess.close()

    @staticmethod
    def is_male(gender_preds):                # facial_analysis.py:76-81, use_sota=False
        return (gender_preds >= 0.6)

    # ---- facial_analysis.py:83-130 -------------------------------------------------------------
    def load_age_gender(self, model_file, max_batch, device):
        graph = read_graph(model_file)
        outs = {OUT_AGE: 'age_pred/Softmax:0', OUT_GENDER: 'gender_pred/Sigmoid:0',
                OUT_FEATURES: 'global_pooling/Mean:0'}
        for t in outs.values():
            graph.get_tensor_by_name(t)       # KeyError like graph.get_tensor_by_name (:84-86)
        in_node, _ = graph.get_tensor_by_name('input_1:0')
        _, w, h, _ = graph.placeholder_shape(in_node.name)
        self.w, self.h = int(w), int(h)
        # the preprocessing of facial_analysis.py:95-107 (uint8 pixels minus the ImageNet-Caffe BGR mean) bounds the input: |x| < 256
        # ... and lets the engine take the RESIZED BYTES themselves (Engine.forward_u8): the float32 conversion, the channel
        # reversal and the float32 mean of :98-107 are folded into the first kernel's constants
        mean32 = tuple(float(np.float32(m)) for m in preprocess.IMAGENET_CAFFE_BGR_MEAN)
        self.plan = lower_graph(graph, 'input_1:0', outs, (self.w, self.h), input_bound=256.0, u8_mean_bgr=mean32)
        # age_gender_fun is called once per face (:93-129, process_image :225-294): few images per call run the small-batch
        # lowering of the same graph (Engine.__init__)
        small = (lower_graph(graph, 'input_1:0', outs, (self.w, self.h), input_bound=256.0, u8_mean_bgr=mean32, presplit="none")
                 if self.latency_plan else None)
        self.sess = Engine(self.plan, max_batch=max_batch, device=device, small_plan=small)

        def age_gender_fun(img):
            ages, genders, feats = self.age_gender_batch([img], latency=self.latency_plan)
            if self.print_stat:
                print('gender', genders[0])
                print('age', ages[0])
            return ages[0], genders[0], feats[0]
        return age_gender_fun

    def preprocess_face(self, img_rgb_u8: np.ndarray) -> np.ndarray:
        """facial_analysis.py:95-107: cv2.resize -> float32 -> BGR -> ImageNet-Caffe mean."""
        resized = preprocess.resize_linear_u8(img_rgb_u8, self.w, self.h)
        return preprocess.to_model_input(resized, True, True, dtype=np.float32)

    def age_gender_batch(self, faces_rgb_u8: Sequence[np.ndarray], latency: bool = False):
        """-> (ages [float], genders [ndarray[1]], features [ndarray[1024]]) for a list of face crops.
        latency: see Engine.forward (a frame's few faces: process_image and age_gender_fun pass it)."""
        torch = _lib.require_gpu()
        ages, genders, feats = [], [], []
        mb = self.sess.max_batch
        for i in range(0, len(faces_rgb_u8), mb):
            chunk = faces_rgb_u8[i:i + mb]
            if self.device_preprocess and self.sess.accepts_u8:      # cv2.resize on the GPU (integer: same bits), bytes into the net
                from . import preprocess_device
                x8 = preprocess_device.preprocess_faces_cv(chunk, (self.h, self.w), device=self.sess.device, raw_u8=True)
                r = self.sess.forward_u8(x8, (OUT_FEATURES, OUT_AGE, OUT_GENDER), latency=latency)
            else:
                if self.device_preprocess:  # cv2.resize + BGR + mean on the GPU
                    from . import preprocess_device
                    xd = preprocess_device.preprocess_faces_cv(chunk, (self.h, self.w), device=self.sess.device)
                else:
                    x = np.stack([self.preprocess_face(f) for f in chunk])
                    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self.sess.device)
                r = self.sess.forward(xd, (OUT_FEATURES, OUT_AGE, OUT_GENDER), latency=latency)
            # one read-back (three small device-to-host copies are three synchronisations)
            packed = torch.cat([r["age_probs"], r["gender"], r["features"]], dim=1).cpu().numpy()
            na, ng = r["age_probs"].shape[1], r["gender"].shape[1]
            age_p, gen, fea = packed[:, :na], packed[:, na:na + ng].copy(), packed[:, na + ng:].copy()
            for j in range(len(chunk)):
                ages.append(decode_age(age_p[j])[0])
                genders.append(gen[j])
                feats.append(fea[j])
        return ages, genders, feats

    # ---- facial_analysis.py:210-223 ---------------------------------------------------------------
    def detect_faces(self, img):
        if self.detector is None:
            raise NotImplementedError("no face detector configured: pass detector=callable(img_rgb) -> "
                                      "(bounding_boxes, points); the MTCNN cascade is the step before this path")
        return self.detector(img)

    # ---- facial_analysis.py:225-294 -----------------------------------------------------------------
    @staticmethod
    def face_boxes(bounding_boxes, img_w: int, img_h: int) -> List[List[int]]:
        """The per-bbox geometry of process_image: int cast, keep x2>x1 and y2>y1, pad 10 px,
        clip to the frame (:233-263)."""
        out = []
        for b in bounding_boxes:
            b = [int(bi) for bi in b]
            x1, y1, x2, y2 = b[0:4]
            if x2 > x1 and y2 > y1:
                dw, dh = 10, 10
                x1, x2 = x1 - dw, x2 + dw
                y1, y2 = y1 - dh, y2 + dh
                box = [x1, y1, x2, y2]
                if box[0] < 0:
                    box[0] = 0
                if box[2] > img_w:
                    box[2] = img_w
                if box[1] < 0:
                    box[1] = 0
                if box[3] > img_h:
                    box[3] = img_h
                out.append(box)
        return out

    def process_image(self, draw, bounding_boxes=None, points=None):
        """draw: BGR uint8 frame, as cv2.imread returns it (:225-226).  Detection runs unless
        ``bounding_boxes`` is supplied.  Returns (bboxes, points, ages, genders, facial_features)."""
        draw = np.asarray(draw)
        # cv2.cvtColor(draw, COLOR_BGR2RGB): plane by plane (np.ascontiguousarray(draw[..., ::-1]) walks a 3-element inner loop with
        # a negative stride: 1.2 ms for a 784x588 frame, a quarter of this call, against 0.3 ms)
        img = np.empty(draw.shape, draw.dtype)
        img[..., 0], img[..., 1], img[..., 2] = draw[..., 2], draw[..., 1], draw[..., 0]
        t = time.time()
        if bounding_boxes is None:
            bounding_boxes, points = self.detect_faces(img)
        elapsed = time.time() - t
        if self.print_stat:
            print('detection elapsed', elapsed)
        img_h, img_w, _ = img.shape
        bboxes = self.face_boxes(bounding_boxes, img_w, img_h)
        crops = [img[y1:y2, x1:x2, :] for (x1, y1, x2, y2) in bboxes]
        t = time.time()
        ages, genders, facial_features = self.age_gender_batch(crops, latency=self.latency_plan) if crops else ([], [], [])
        if self.print_stat:
            print('age gender elapsed', time.time() - t)
        return bboxes, points if points is not None else [], ages, genders, facial_features
