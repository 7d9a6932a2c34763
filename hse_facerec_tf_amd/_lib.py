"""ctypes binding of libhsefr.so (include/hsefr.h).  No fallback: if the library is missing the
import of anything that needs it raises, and every non-zero status becomes a Python exception
in the reference's error style (ValueError for shape errors, NotImplementedError for unsupported
graphs, RuntimeError otherwise)."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# HSEFR_LIB selects another build of the library (e.g. libhsefr_dev.so for tools/kbench.py); the product is libhsefr.so
LIB_PATH = os.path.join(_HERE, os.environ.get("HSEFR_LIB", "libhsefr.so"))

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM, ERR_SHAPE = 0, -1, -2, -3, -4, -5

_lib = None

# name -> (restype, argtypes); one row per declaration in include/hsefr.h
_fp = c_void_p   # device pointers travel as integers (tensor.data_ptr())
SIGNATURES = {
    "hsefr_version": (c_int, []),
    "hsefr_last_error_string": (c_char_p, []),
    "hsefr_engine_set_graph_batch": (c_int, [c_void_p, c_int]),
    "hsefr_engine_graph_launches": (c_longlong, [c_void_p]),
    "hsefr_engine_create": (c_int, [c_void_p, c_size_t, c_int, POINTER(c_void_p)]),
    "hsefr_plan_validate": (c_int, [c_void_p, c_size_t]),
    "hsefr_plan_describe": (c_int, [c_void_p, c_size_t, c_int, c_char_p, c_size_t]),
    "hsefr_engine_workspace_bytes": (c_size_t, [c_void_p]),
    "hsefr_engine_max_batch": (c_int, [c_void_p]),
    "hsefr_engine_forward": (c_int, [c_void_p, _fp, c_int, _fp, _fp, _fp, c_void_p]),
    "hsefr_engine_buffer": (c_void_p, [c_void_p, c_int]),
    "hsefr_engine_copy_buffer": (c_int, [c_void_p, c_int, _fp, c_size_t, c_void_p]),
    "hsefr_engine_set_profiling": (c_int, [c_void_p, c_int]),
    "hsefr_engine_profiled_calls": (c_longlong, [c_void_p]),
    "hsefr_engine_op_times_ms": (c_int, [c_void_p, c_int, POINTER(c_float), c_int]),
    "hsefr_engine_destroy": (c_int, [c_void_p]),
    "hsefr_conv_c3_bias_act": (c_int, [_fp, _fp, _fp, _fp] + [c_int] * 12 + [c_void_p]),
    "hsefr_dwconv3x3_bn_relu6": (c_int, [_fp, _fp, _fp, _fp, _fp] + [c_int] * 10 + [c_void_p]),
    "hsefr_dwconv3x3_bn_relu6_split": (c_int, [_fp, _fp, _fp, _fp, _fp] + [c_int] * 11 + [c_void_p]),
    "hsefr_pwconv1x1_presplit_gap": (c_int, [_fp, c_void_p, _fp, _fp, _fp, c_longlong, c_int, c_int, c_int, c_int, c_void_p]),
    "hsefr_pwconv1x1_presplit_dw": (c_int, [_fp, c_void_p, _fp, _fp, _fp, _fp, c_longlong] + [c_int] * 7 + [c_void_p]),
    "hsefr_pwconv1x1_presplit": (c_int, [_fp, c_void_p, _fp, _fp, _fp, c_longlong, c_int, c_int, c_int, c_void_p]),
    "hsefr_pwconv1x1_bias_relu6": (c_int, [_fp, _fp, _fp, _fp, c_longlong, c_int, c_int, c_int, c_void_p]),
    "hsefr_stem2_fused": (c_int, [_fp] * 6 + [c_void_p] + [_fp] * 6 + [c_int] * 13 + [c_void_p]),
    "hsefr_stem3_fused": (c_int, [_fp, c_void_p] + [_fp] * 5 + [c_void_p] + [_fp] * 7 + [c_int] * 14 + [c_void_p]),
    "hsefr_stem4_fused": (c_int, [_fp, c_int, c_void_p] + [_fp] * 5 + [c_void_p] + [_fp] * 7 + [c_int] * 6 + [c_void_p]),
    "hsefr_stem5_stream": (c_int, [_fp, c_int, c_void_p] + [_fp] * 5 + [c_void_p] + [_fp] * 7 + [c_int] * 6 + [c_void_p]),
    "hsefr_engine_accepts_u8": (c_int, [c_void_p]),
    "hsefr_engine_forward_u8": (c_int, [c_void_p, _fp, c_int, _fp, _fp, _fp, c_void_p]),
    "hsefr_engine_input_overflow": (c_int, [c_void_p, POINTER(c_int), c_void_p]),
    "hsefr_engine_input_overflow_async": (c_int, [c_void_p, POINTER(c_int), c_void_p]),
    "hsefr_dwpw_f16split": (c_int, [_fp, _fp, _fp, _fp, c_void_p, _fp, _fp, _fp] + [c_int] * 12 + [c_void_p]),
    "hsefr_pwconv1x1_f16split": (c_int, [_fp, c_void_p, _fp, _fp, _fp, c_longlong, c_int, c_int, c_int, c_int, c_void_p]),
    "hsefr_dwpw_fused": (c_int, [_fp] * 7 + [c_int] * 10 + [c_void_p]),
    "hsefr_gap": (c_int, [_fp, _fp, c_int, c_int, c_int, c_void_p]),
    "hsefr_dense": (c_int, [_fp, _fp, _fp, _fp, c_int, c_int, c_int, c_int, c_void_p]),
    "hsefr_softmax": (c_int, [_fp, _fp, c_int, c_int, c_void_p]),
    "hsefr_heads_fused": (c_int, [_fp] * 11 + [c_int] * 3 + [c_void_p]),
    "hsefr_conv_bf16": (c_int, [_fp, _fp, _fp, _fp, _fp, _fp] + [c_int] * 13 + [c_void_p]),
    "hsefr_conv1x1_proj_bf16": (c_int, [_fp] * 9 + [c_int] * 10 + [c_void_p]),
    "hsefr_conv1x1_sres_bf16": (c_int, [_fp] * 6 + [c_int] * 9 + [c_void_p]),
    "hsefr_conv1x1_pair_bf16": (c_int, [_fp] * 14 + [c_longlong] + [c_int] * 6 + [c_void_p]),
    "hsefr_conv1x1_pair_sub2_bf16": (c_int, [_fp] * 14 + [c_int] * 9 + [c_void_p]),
    "hsefr_stem7x7_bf16": (c_int, [_fp, _fp, _fp, _fp, _fp] + [c_int] * 6 + [c_void_p]),
    "hsefr_stem7x7_pool_bf16": (c_int, [_fp, c_void_p, _fp, _fp, c_void_p] + [c_int] * 7 + [c_void_p]),
    "hsefr_maxpool3x3s2_bf16": (c_int, [_fp, _fp] + [c_int] * 8 + [c_void_p]),
    "hsefr_gap_bf16": (c_int, [_fp, _fp, c_int, c_int, c_int, c_void_p]),
    "hsefr_preprocess_pil_u8": (c_int, [_fp, _fp, _fp] + [c_int] * 5 + [_fp, _fp, _fp, c_int, _fp, _fp, _fp, c_int, c_int,
                                        POINTER(ctypes.c_double), c_void_p]),
    "hsefr_preprocess_cv_u8": (c_int, [_fp, _fp] + [c_int] * 5 + [_fp] * 6 + [c_int, POINTER(ctypes.c_double), c_void_p]),
    "hsefr_l2_normalize": (c_int, [_fp, _fp, c_int, c_int, c_void_p]),
    "hsefr_conv2d_direct": (c_int, [_fp] * 5 + [c_int] * 12 + [c_void_p]),
    "hsefr_conv2d_f32": (c_int, [_fp] * 6 + [c_int] * 13 + [c_void_p]),
    "hsefr_conv2d_f32_mfma": (c_int, [_fp] * 6 + [c_int] * 13 + [c_void_p]),
    "hsefr_maxpool_f32": (c_int, [_fp, _fp] + [c_int] * 10 + [c_void_p]),
    "hsefr_mtcnn_pyramid_level": (c_int, [_fp, _fp, c_int, c_int, c_int, c_int, c_void_p]),
    "hsefr_mtcnn_post_capacity": (c_int, []),
    "hsefr_mtcnn_stage1_level": (c_int, [_fp, _fp, c_int, c_int, ctypes.c_double, ctypes.c_float, _fp, _fp, c_void_p]),
    "hsefr_mtcnn_stage1_finish": (c_int, [_fp, _fp, _fp, _fp, c_int, c_int, c_void_p]),
    "hsefr_mtcnn_stage_finish": (c_int, [c_int, _fp, c_int, _fp, _fp, _fp, ctypes.c_float, _fp, _fp, _fp, _fp, c_int, c_int, c_void_p]),
    "hsefr_mtcnn_nms": (c_int, [_fp, c_int, ctypes.c_double, c_int, _fp, _fp, c_void_p]),
    "hsefr_mtcnn_crops": (c_int, [_fp, _fp, _fp, c_int, c_int, c_int, c_int, c_void_p]),
    "hsefr_pairwise_dist": (c_int, [_fp, _fp, c_int, c_int, c_int, _fp, c_void_p]),
    "hsefr_nn1": (c_int, [_fp, _fp, c_int, c_int, c_int, _fp, _fp, c_void_p]),
    "hsefr_nn1_fallbacks": (c_longlong, []),
}

# development builds only (csrc/hsefr_dev.h; build.sh with HSEFR_DEV=1): bound when the loaded library has them
DEV_SIGNATURES = {
    "hsefr_debug_set": (c_int, [c_char_p, c_int]),
    "hsefr_debug_read_stamps": (c_int, [c_int, c_void_p, ctypes.c_size_t]),
    "hsefr_debug_clock_probe": (c_int, [_fp, c_int, c_int, c_void_p]),
    "hsefr_debug_copy": (c_int, [_fp, _fp, c_size_t, c_void_p]),
    "hsefr_stem_fused": (c_int, [_fp] * 6 + [c_void_p, _fp, _fp, _fp] + [c_int] * 9 + [c_void_p]),     # round 1's stem: dev builds only since round 6
}


class HsefrError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load libhsefr.so once.  torch is imported first on purpose: the library's DT_NEEDED
    libamdhip64.so.7 must resolve to the HIP runtime PyTorch already mapped, so that torch
    streams and ``data_ptr()`` addresses belong to the same runtime instance."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (side effect: maps torch's libamdhip64)
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in DEV_SIGNATURES.items():
            if hasattr(L, name):
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    s = lib().hsefr_last_error_string()
    return s.decode("utf-8", "replace") if s else ""


def check(status: int, what: str = "") -> None:
    if status == OK:
        return
    msg = "%s%s (hsefr status %d)" % (what + ": " if what else "", last_error(), status)
    if status == ERR_SHAPE or status == ERR_INVALID:
        raise ValueError(msg)
    if status == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if status == ERR_NOMEM:
        raise MemoryError(msg)
    raise HsefrError(msg)


_torch_gpu = None      # torch, once a GPU has been seen (torch.cuda.is_available() re-reads the environment on every call: the
                       # small-net paths make ~100 launches per frame and spent a tenth of their host time asking)


def require_gpu():
    """The product path runs on an MI355X or not at all."""
    global _torch_gpu
    if _torch_gpu is None:
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("hse_facerec_tf_amd needs a ROCm GPU (gfx950); torch.cuda.is_available() is False "
                               "and there is no CPU fallback")
        _torch_gpu = torch
    return _torch_gpu


def current_stream_ptr() -> int:
    """The current torch stream OF THE CURRENT DEVICE: callers launch under ``on_device(tensor)`` so that the
    current device is the one that owns the pointers they pass."""
    import torch
    try:        # the raw handle without building a torch.cuda.Stream object (what torch's own compiled code paths call)
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:
        return torch.cuda.current_stream().cuda_stream


def cuda_device(device=None):
    """None -> the current device; an index, 'cuda:1' or torch.device -> torch.device('cuda', index)."""
    torch = require_gpu()
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    d = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    if d.type != "cuda":
        raise ValueError("hse_facerec_tf_amd runs on a ROCm GPU, not on %r" % (device,))
    return torch.device("cuda", torch.cuda.current_device() if d.index is None else d.index)


def on_device(t):
    """Context manager: make the device that owns tensor ``t`` (or the given torch.device) current, so that the
    stream handed to libhsefr and the HIP launch target match the pointers (a launch under device 0
    on device-1 pointers is a memory fault, not an exception)."""
    import torch
    dev = t.device if hasattr(t, "device") else t
    try:        # already current (the usual case): nothing to switch, nothing to restore
        if isinstance(dev, torch.device) and dev.index is not None and dev.index == torch._C._cuda_getDevice():
            return _NO_SWITCH
    except AttributeError:
        pass
    return torch.cuda.device(dev)


class _NoSwitch:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()
