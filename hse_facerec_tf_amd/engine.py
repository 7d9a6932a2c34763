"""Python handle on an hsefr engine (the object that stands where ``tf.Session`` stood:
facerec_test.py:58, facial_analysis.py:58).  torch is used for device memory and streams
only: tensors are containers whose ``data_ptr()`` is handed to the C ABI."""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .lowering import OUT_AGE, OUT_FEATURES, OUT_GENDER, Plan

_SLOT_NAMES = {OUT_FEATURES: "features", OUT_AGE: "age_probs", OUT_GENDER: "gender"}


SMALL_BATCH = 32      # forward(..., latency=True) of at most this many images runs the small-batch plan, when the caller lowered one
BULK_CHUNK = 1024     # a forward of more images than this goes through the network in chunks of this many (Engine.__init__)


class Engine:
    def __init__(self, plan: Plan, max_batch: int = 256, device: Optional[int] = None, small_plan: Optional[Plan] = None,
                 small_batch: int = SMALL_BATCH, bulk_chunk: int = BULK_CHUNK):
        """bulk_chunk: forwards of more images than this run chunk by chunk through ALL layers (same stream, same bits: an image's
        result does not depend on its batch) -- consecutive layers then meet their input in the L2 / Infinity Cache instead of
        HBM (6 144 images of 192 x 192 in one piece: 207 k faces/s; in chunks of 512-2048: 244-245 k), and the activation workspace
        is sized for one chunk (max_batch 6 144: 1.2 GB instead of 14).  0 = never chunk.
        small_plan: a second lowering of the SAME graph for forwards of at most small_batch images -- the reference calls
        its session once per image (facerec_test.py:114-122, facial_analysis.py:93-129), and a plan tuned for 256 images per launch
        (GEMM tiles of 288 rows with the next depthwise in their epilogue: one tile, 4 of 256 CUs, 28 us per layer whatever the
        batch) is not the fastest one for that: lowered with presplit='none' (plain GEMMs + standalone depthwise kernels) one
        image takes 0.17 ms instead of 0.26, 32 images 0.34 instead of 0.36, and from 48 up the default plan wins.  Both plans
        are fp32-grade restatements of the same graph (each is tested against the oracle at the same tolerance); their results
        differ by summation order (3e-7 relative) -- which is why the choice is the CALLER's, per call (latency=True), and never
        made from the batch size alone: the bulk paths (extract_batch / extract_files / config 5's shards) promise that an image's
        embedding does not depend on how the images were batched, and keep the one plan for every batch size."""
        torch = _lib.require_gpu()
        self._torch = torch
        self.plan = plan
        self.device = _lib.cuda_device(device)
        self.max_batch = int(max_batch)
        self.chunk = min(self.max_batch, int(bulk_chunk)) if bulk_chunk and bulk_chunk > 0 else self.max_batch
        self._h = self._create(plan, self.chunk)
        self.in_hwc = plan.in_hwc
        self.out_elems = {slot: elems for slot, (_, elems) in plan.outputs.items()}
        self.small_plan, self.small_batch, self._hs, self._last = None, 0, None, None
        if small_plan is not None and small_batch > 0:
            if small_plan.in_hwc != plan.in_hwc or {s_: e for s_, (_, e) in small_plan.outputs.items()} != self.out_elems:
                raise ValueError("small_plan must be a lowering of the same graph: same input and outputs")
            self.small_plan, self.small_batch = small_plan, min(int(small_batch), self.max_batch)
            self._hs = self._create(small_plan, self.small_batch)

    def _create(self, plan: Plan, max_batch: int):
        blob = plan.serialize()
        handle = ctypes.c_void_p()
        with self._torch.cuda.device(self.device):
            buf = ctypes.create_string_buffer(blob, len(blob))
            _lib.check(_lib.lib().hsefr_engine_create(ctypes.cast(buf, ctypes.c_void_p), len(blob), int(max_batch),
                                                      ctypes.byref(handle)), "hsefr_engine_create")
        return handle

    def _handle_for(self, n: int, latency: bool):
        self._last = self._hs if latency and self._hs is not None and n <= self.small_batch else self._h
        return self._last

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_hs", None):
            _lib.lib().hsefr_engine_destroy(self._hs)
            self._hs = None
        if getattr(self, "_h", None):
            _lib.lib().hsefr_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self) -> int:
        return int(_lib.lib().hsefr_engine_workspace_bytes(self._h)) + (int(_lib.lib().hsefr_engine_workspace_bytes(self._hs)) if self._hs else 0)

    # -- forward -----------------------------------------------------------------------------
    def forward(self, x, want: Sequence[int] = (OUT_FEATURES,), latency: bool = False) -> Dict[str, "object"]:
        """x: CUDA float32 tensor [n, h, w, c] NHWC contiguous (already preprocessed).  Returns
        {'features'|'age_probs'|'gender': CUDA tensor}.  Asynchronous on the current stream.
        latency=True: a call of the reference's one-image-per-run kind -- at most small_batch images take the small-batch
        plan if the engine has one (__init__)."""
        torch = self._torch
        if self._h is None:
            raise RuntimeError("Attempted to use a closed Session.")   # TF's message for a closed session
        h, w, c = self.in_hwc
        if x.dim() != 4 or tuple(x.shape[1:]) != (h, w, c):
            raise ValueError("Cannot feed value of shape %r for Tensor which has shape '(?, %d, %d, %d)'"
                             % (tuple(x.shape), h, w, c))
        if x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous():
            raise ValueError("engine input must be a contiguous float32 CUDA tensor")
        if x.device != self.device:      # weights and activation buffers live on self.device: never launch across devices
            raise ValueError("engine input is on %s but the engine was created on %s" % (x.device, self.device))
        n = int(x.shape[0])
        outs = {}
        ptrs = [None, None, None]
        for slot in want:
            if slot not in self.out_elems:
                raise KeyError("the plan has no output %r" % _SLOT_NAMES[slot])
            t = torch.empty((n, self.out_elems[slot]), dtype=torch.float32, device=x.device)
            outs[_SLOT_NAMES[slot]] = t
            ptrs[slot] = t.data_ptr()
        self._run("hsefr_engine_forward", x.data_ptr(), 4 * h * w * c, n, ptrs, latency)
        return outs

    def _run(self, entry: str, x_ptr: int, image_bytes: int, n: int, ptrs, latency: bool) -> None:
        """One C forward, or one per chunk of self.chunk images (pointers advanced by whole images; same stream)."""
        if n > self.max_batch:
            raise ValueError("forward: batch %d outside [0, %d]" % (n, self.max_batch))
        fn = getattr(_lib.lib(), entry)
        with self._torch.cuda.device(self.device):
            stream = _lib.current_stream_ptr()
            if n <= self.chunk:
                _lib.check(fn(self._handle_for(n, latency), x_ptr, n, ptrs[0], ptrs[1], ptrs[2], stream), entry)
                return
            for a in range(0, n, self.chunk):
                m = min(self.chunk, n - a)
                sub = [None if p_ is None else p_ + 4 * a * self.out_elems[slot] for slot, p_ in enumerate(ptrs)]
                _lib.check(fn(self._handle_for(m, False), x_ptr + a * image_bytes, m, sub[0], sub[1], sub[2], stream), entry)

    @property
    def accepts_u8(self) -> bool:
        """True if forward_u8 works on this plan (fused stem lowered with a BGR mean, input edges multiples of 4)."""
        return bool(self._h) and bool(_lib.lib().hsefr_engine_accepts_u8(self._h))

    def forward_u8(self, x_u8, want: Sequence[int] = (OUT_FEATURES,), latency: bool = False) -> Dict[str, "object"]:
        """forward() on the RESIZED image bytes: x_u8 CUDA uint8 [n, h, w, 3], RGB as the decoder / resizer left them.  Float
        conversion, channel reversal and mean subtraction (facerec_test.py:95-106) happen inside the first kernel."""
        torch = self._torch
        if self._h is None:
            raise RuntimeError("Attempted to use a closed Session.")
        h, w, c = self.in_hwc
        if x_u8.dim() != 4 or tuple(x_u8.shape[1:]) != (h, w, c):
            raise ValueError("Cannot feed value of shape %r for Tensor which has shape '(?, %d, %d, %d)'" % (tuple(x_u8.shape), h, w, c))
        if x_u8.dtype != torch.uint8 or not x_u8.is_cuda or not x_u8.is_contiguous():
            raise ValueError("forward_u8 takes a contiguous uint8 CUDA tensor")
        if x_u8.device != self.device:
            raise ValueError("engine input is on %s but the engine was created on %s" % (x_u8.device, self.device))
        n = int(x_u8.shape[0])
        outs, ptrs = {}, [None, None, None]
        for slot in want:
            if slot not in self.out_elems:
                raise KeyError("the plan has no output %r" % _SLOT_NAMES[slot])
            t = torch.empty((n, self.out_elems[slot]), dtype=torch.float32, device=x_u8.device)
            outs[_SLOT_NAMES[slot]] = t
            ptrs[slot] = t.data_ptr()
        self._run("hsefr_engine_forward_u8", x_u8.data_ptr(), h * w * c, n, ptrs, latency)
        return outs

    def forward_all_layers(self, x) -> None:
        """Run every op of the BULK plan (no fetch pruning); intermediate buffers can then be read with layer_output.  A debug
        entry: the activation workspace holds one chunk (bulk_chunk), so at most `self.chunk` images -- more would need the
        layer buffers of several chunks at once."""
        if x.device != self.device:
            raise ValueError("engine input is on %s but the engine was created on %s" % (x.device, self.device))
        if int(x.shape[0]) > self.chunk:
            raise ValueError("forward_all_layers: batch %d outside [0, %d] (the workspace holds one chunk of bulk_chunk images; "
                             "create the engine with a larger bulk_chunk to inspect more)" % (int(x.shape[0]), self.chunk))
        with self._torch.cuda.device(self.device):
            _lib.check(_lib.lib().hsefr_engine_forward(self._h, x.data_ptr(), int(x.shape[0]), None, None, None,
                                                       _lib.current_stream_ptr()), "hsefr_engine_forward")

    def layer_output(self, layer_index: int, n: int):
        """Copy of a layer's activation buffer as a CUDA tensor [n, oh, ow, c].  Only meaningful
        right after the op ran and before a later op recycled the buffer.  A layer whose result is stored as split
        rows for the GEMM behind it (Layer.out_split) is decoded back to fp32 values."""
        torch = self._torch
        if not 0 <= int(n) <= self.chunk:
            raise ValueError("layer_output: n = %d outside [0, %d] (one chunk of bulk_chunk images)" % (int(n), self.chunk))
        L = self.plan.layers[layer_index]
        out = torch.empty((n,) + tuple(L.out_shape), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hsefr_engine_copy_buffer(self._h, L.out_buf, out.data_ptr(), out.numel() * 4,
                                                           _lib.current_stream_ptr()), "hsefr_engine_copy_buffer")
        if getattr(L, "out_split", 0):
            from . import ops
            c = L.out_shape[2]
            rows = out.view(torch.float16).reshape((n,) + tuple(L.out_shape[:2]) + (c // 32, 2, 32))
            return ops.split_rows_decode(rows, L.out_split)
        return out

    def input_overflow(self) -> bool:
        """True if any forward since the last call fed a value outside the plan's declared input bound (lower_graph
        input_bound; the results of such a forward are meaningless).  Synchronises the current stream and clears the flag."""
        any_set = False
        with self._torch.cuda.device(self.device):
            for h in (self._h, self._hs):
                if h:
                    flag = ctypes.c_int(0)
                    _lib.check(_lib.lib().hsefr_engine_input_overflow(h, ctypes.byref(flag), _lib.current_stream_ptr()),
                               "hsefr_engine_input_overflow")
                    any_set = any_set or bool(flag.value)
        return any_set

    def input_overflow_async(self, pinned_host_int_ptr: int) -> None:
        """The same read-and-clear, asynchronous: the flag lands in a PINNED host int32 (address given) when the current
        stream reaches this point; the caller waits on an event of its own before reading it.  Nothing here waits for the device
        (hsefr_engine_input_overflow_async only enqueues the copy and the clear)."""
        # (the flag of the engine the LAST forward ran on: callers pair every forward with one read, and a forward runs on one
        # handle only -- chunked forwards never take the small plan; input_overflow() reads both.  A flag left on the OTHER handle
        # by an unpaired earlier forward stays there until that handle's next read: it is never lost, only late)
        with self._torch.cuda.device(self.device):
            _lib.check(_lib.lib().hsefr_engine_input_overflow_async(self._last or self._h, ctypes.cast(ctypes.c_void_p(pinned_host_int_ptr), ctypes.POINTER(ctypes.c_int)),
                                                                    _lib.current_stream_ptr()), "hsefr_engine_input_overflow_async")

    # -- small batches as one hipGraph launch ---------------------------------------------------
    def set_graph_batch(self, max_n: int) -> None:
        """Forwards of at most max_n images replay a captured hipGraph (default 0 = off: no faster on the device).  The setting
        goes to BOTH plans (each capped at what its handle was created for: one chunk / small_batch), so a forward replays
        graphs whichever plan its `latency` flag selects."""
        max_n = int(max_n)
        if max_n < 0:
            raise ValueError("set_graph_batch: max_n = %d" % max_n)
        _lib.check(_lib.lib().hsefr_engine_set_graph_batch(self._h, min(max_n, self.chunk)))
        if self._hs:
            _lib.check(_lib.lib().hsefr_engine_set_graph_batch(self._hs, min(max_n, self.small_batch)))

    def graph_launches(self) -> int:
        """Graph replays so far, both plans together."""
        return int(_lib.lib().hsefr_engine_graph_launches(self._h)) + (int(_lib.lib().hsefr_engine_graph_launches(self._hs)) if self._hs else 0)

    # -- profiling -----------------------------------------------------------------------------
    def set_profiling(self, depth: int) -> None:
        """depth > 0: keep HIP-event timings of the last `depth` forwards (ring) of EACH plan; 0: off.  A forward of more than
        bulk_chunk images is several C forwards: one ring slot per chunk."""
        _lib.check(_lib.lib().hsefr_engine_set_profiling(self._h, int(depth)))
        if self._hs:
            _lib.check(_lib.lib().hsefr_engine_set_profiling(self._hs, int(depth)))

    def profiled_calls(self, small: bool = False) -> int:
        return int(_lib.lib().hsefr_engine_profiled_calls(self._hs if small and self._hs else self._h))

    def op_times_ms(self, slot: int = 0, small: bool = False) -> List[float]:
        """Per-op times of ring slot `slot`: of the bulk plan, or (small=True) of the small-batch plan's own layer list."""
        h, plan = (self._hs, self.small_plan) if small and self._hs else (self._h, self.plan)
        n = len(plan.layers)
        arr = (ctypes.c_float * n)()
        _lib.check(_lib.lib().hsefr_engine_op_times_ms(h, int(slot), arr, n))
        return list(arr)
