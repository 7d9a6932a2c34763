"""CPU suite, part 3: the C-ABI library loads without a GPU and exports exactly what
include/hsefr.h declares (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hsefr.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hsefr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_documented_surface():
    names = declared_functions()
    for must in ("hsefr_version", "hsefr_engine_create", "hsefr_engine_forward", "hsefr_engine_destroy",
                 "hsefr_last_error_string", "hsefr_dwconv3x3_bn_relu6", "hsefr_pwconv1x1_bias_relu6",
                 "hsefr_conv_c3_bias_act", "hsefr_gap", "hsefr_dense", "hsefr_nn1"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from hse_facerec_tf_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = _lib.lib()
    for name in declared_functions():
        assert hasattr(L, name), "libhsefr.so does not export %s" % name
    # and the binding table covers the header one-to-one
    assert sorted(_lib.SIGNATURES) == declared_functions()
    assert L.hsefr_version() == 141


def test_code_object_targets_gfx950_only():
    from hse_facerec_tf_amd import _lib
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", _lib.LIB_PATH], capture_output=True, text=True)
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"gfx1100"):
        assert other not in blob


def test_argument_validation_needs_no_gpu():
    """Status codes and error strings work before any device is touched."""
    from hse_facerec_tf_amd import _lib
    L = _lib.lib()
    h = ctypes.c_void_p()
    rc = L.hsefr_engine_create(None, 0, 1, ctypes.byref(h))
    assert rc == _lib.ERR_INVALID and "null" in _lib.last_error()
    junk = ctypes.create_string_buffer(b"\0" * 128, 128)
    rc = L.hsefr_engine_create(ctypes.cast(junk, ctypes.c_void_p), 128, 4, ctypes.byref(h))
    assert rc == _lib.ERR_INVALID and "magic" in _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "hsefr_engine_create")
    assert L.hsefr_engine_destroy(None) == 0
    # shape checks of the fused-block entry points come before any device work too
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.POINTER(ctypes.c_float))
    vp = ctypes.cast(buf, ctypes.c_void_p)
    rc = L.hsefr_dwpw_f16split(p, p, p, p, vp, p, p, p, 1, 8, 8, 128, 1, 1, 1, 8, 8, 128, 13, 2, None)                 # a_log2 out of range
    assert rc == _lib.ERR_INVALID and "a_log2" in _lib.last_error()
    rc = L.hsefr_dwpw_f16split(p, p, p, p, vp, p, p, p, 1, 8, 8, 48, 1, 1, 1, 8, 8, 64, 12, 2, None)                   # c % 32 != 0
    assert rc == _lib.ERR_UNSUPPORTED


def test_no_unguarded_store_data_hazard_in_the_device_code():
    """gfx950 needs two instructions between a 16-byte store and a vector write of its data registers (one for the
    SGPR-soffset buffer form); hipcc guarantees one (none).  tools/store_hazard_probe.hip measured it
    (profiles/r01_store_hazard_probe.txt); tools/isa_lint.py disassembles the shipped library and must find no such pair,
    and must still recognise one in a hand-made listing."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump on this machine")
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    from hse_facerec_tf_amd import _lib
    findings, n_symbols, n_stores = lint.lint(_lib.LIB_PATH)
    assert n_symbols > 50 and n_stores > 100, "the lint did not see the device code"
    assert findings == []
    listing = """
0000000000001000 <kernel_a>:
	buffer_store_dwordx4 v[0:3], v112, s[4:7], s70 offen       // 0000
	s_nop 0                                                    // 0008
	v_pk_fma_f32 v[2:3], v[10:11], v[34:35], v[38:39]          // 000C
	global_store_dwordx4 v[8:9], v[4:7], off                   // 0014
	s_nop 1                                                    // 001C
	v_mov_b32_e32 v7, 0                                        // 0020
	global_store_dwordx4 v[8:9], v[4:7], off                   // 0024
	v_mov_b32_e32 v8, 0                                        // 002C
"""
    found = lint.scan_listing(listing)
    assert len(found) == 1 and found[0][3] == [("v", 2), ("v", 3)] and found[0][4] == 1
    # a wide store followed by a TAKEN branch is checked against the first instructions of the branch target
    branchy = """
0000000000002000 <kernel_b>:
	buffer_store_dwordx4 v[0:3], v112, s[4:7], 0 offen        // 0000
	s_cbranch_scc1 L42                                         // 0008
	s_nop 1                                                    // 000C
	v_mov_b32_e32 v9, 0                                        // 0010
0000000000002014 <L42>:
	v_pk_mul_f32 v[0:1], v[10:11], v[34:35]                    // 0014
	s_endpgm                                                   // 001C
"""
    found = lint.scan_listing(branchy)
    assert len(found) == 1 and found[0][3] == [("v", 0), ("v", 1)]
    # the form llvm-objdump really prints: numeric simm16 offsets (in dwords, from the next instruction) + address comments
    numeric = """
0000000000003000 <kernel_c>:
	v_pk_mul_f32 v[2:3], v[10:11], v[34:35]                    // 000000003000: D3B10002 1802450A
	s_nop 0                                                    // 000000003008: BF800000
	global_store_dwordx4 v[8:9], v[0:3], off                   // 00000000300C: DC7C8000 007F0008
	s_cbranch_scc1 65530                                       // 000000003014: BF85FFFA <kernel_c+0x0>
	s_endpgm                                                   // 000000003018: BF810000
"""
    found = lint.scan_listing(numeric)
    assert len(found) == 1 and found[0][3] == [("v", 2), ("v", 3)]
    # second rule: the 16-bit-writing pack instruction hipcc treats as a full-register write (profiles/r03_ashr_pk_probe.txt)
    half = """
0000000000004000 <kernel_d>:
	v_ashr_pk_u8_i32 v5, v5, v7, 22                            // 000000004000: D2650005 025A0F05
	v_or3_b32 v5, v5, v6, v1                                   // 000000004008: D1FF0005 04060D05
	s_endpgm                                                   // 000000004010: BF810000
"""
    found = lint.scan_listing(half)
    assert len(found) == 1 and "v_ashr_pk_u8_i32" in found[0][1]
    # and a library in which nothing could be scanned does not pass vacuously
    import subprocess, sys, tempfile
    with tempfile.NamedTemporaryFile(suffix=".so") as f:
        f.write(b"\x7fELF" + b"\0" * 60)
        f.flush()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"), f.name], capture_output=True, text=True)
    assert r.returncode != 0


def test_product_library_exports_the_documented_surface_only():
    """No tuning knobs, calibration kernels or C++ internals in the product ABI: the dynamic symbol
    table holds exactly the functions include/hsefr.h declares."""
    from hse_facerec_tf_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    syms = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert [x for x in syms if "debug" in x] == []
    extra = [x for x in syms if x not in declared_functions() and not x.startswith(("_init", "_fini"))]
    assert extra == [], extra


def test_plan_validation_guards_the_in_place_outputs():
    """forward() lets the producing kernel write the caller's [n, out_elems] tensor in place of the plan's output
    buffer, so engine_create must reject a plan whose output buffer has two producers or another size (validated before any
    device is touched)."""
    import copy
    from conftest import MODEL_PB
    from hse_facerec_tf_amd import _lib, graphdef, lowering
    L = _lib.lib()
    plan = lowering.lower_graph(graphdef.read_graph(MODEL_PB), "input_1:0", {0: "global_pooling/Mean:0", 1: "age_pred/Softmax:0",
                                                                             2: "gender_pred/Sigmoid:0"}, (96, 96), launch_fusion=False)

    def create(p):
        blob = p.serialize()
        buf = ctypes.create_string_buffer(blob, len(blob))
        h = ctypes.c_void_p()
        rc = L.hsefr_engine_create(ctypes.cast(buf, ctypes.c_void_p), len(blob), 2, ctypes.byref(h))
        if rc == 0:
            L.hsefr_engine_destroy(h)
        return rc, _lib.last_error()
    out_layer = plan.outputs[0][0]
    out_buf = plan.layers[out_layer].out_buf
    # (a) a second op writing the output buffer
    bad = copy.deepcopy(plan)
    victim = next(i for i, l in enumerate(bad.layers) if i != out_layer and l.out_buf != out_buf
                  and l.out_bytes <= bad.buffers[out_buf])
    bad.layers[victim].out_buf = out_buf
    rc, msg = create(bad)
    assert rc == _lib.ERR_INVALID and "producing ops" in msg, (rc, msg)
    # (b) a slot that declares fewer elements than its buffer holds
    bad = copy.deepcopy(plan)
    bad.outputs[0] = (out_layer, plan.outputs[0][1] // 2)
    rc, msg = create(bad)
    assert rc == _lib.ERR_INVALID and "the slot declares" in msg, (rc, msg)
    # the untouched plan passes validation (and then fails, or not, only for lack of a GPU)
    rc, msg = create(plan)
    assert rc == 0 or "producing ops" not in msg and "the slot declares" not in msg


def test_plan_validate_alone_and_the_launch_fusion_flags():
    """Round 6: hsefr_plan_validate runs engine_create's pre-device checks alone, and the fusion flags of hsefr_plan_op are validated
    against the pattern their fused launch computes -- a flag on the wrong op, a second output that aliases an operand, a pair whose
    shapes the kernel does not cover and an unknown flag bit are all refused before any device is touched."""
    import copy
    from conftest import MODEL_PB
    from hse_facerec_tf_amd import _lib, graphdef, lowering, resnet50
    L = _lib.lib()

    def validate(p):
        blob = p.serialize()
        buf = ctypes.create_string_buffer(blob, len(blob))
        rc = L.hsefr_plan_validate(ctypes.cast(buf, ctypes.c_void_p), len(blob))
        return rc, _lib.last_error()
    rn = resnet50.build_plan(resnet50.synthetic_weights(1), (64, 64), "caffe")
    pairs = [i for i, x in enumerate(rn.layers) if x.flags & lowering.OPF_PAIR_NEXT]
    assert [rn.layers[i].name for i in pairs] == ["conv2_1_1x1_increase", "conv2_2_1x1_increase"]
    assert validate(rn)[0] == 0
    bad = copy.deepcopy(rn)                                           # the last increase layer of the stage: the op behind it has stride 2
    bad.layers[pairs[1] + 3].flags = lowering.OPF_PAIR_NEXT
    rc, msg = validate(bad)
    assert rc == _lib.ERR_INVALID and "PAIR_NEXT" in msg, (rc, msg)
    bad = copy.deepcopy(rn)                                           # the reduce layer's output in the increase layer's input buffer
    bad.layers[pairs[0] + 1].out_buf = rn.layers[rn.layers[pairs[0]].src].out_buf
    rc, msg = validate(bad)
    assert rc == _lib.ERR_INVALID and ("aliases" in msg or "exceeds" in msg), (rc, msg)
    bad = copy.deepcopy(rn)                                           # a stage-3 increase layer (128 -> 512): pattern fine up to the shapes
    j = next(i for i, x in enumerate(bad.layers) if x.name == "conv3_2_1x1_increase")
    bad.layers[j].flags = lowering.OPF_PAIR_NEXT
    rc, msg = validate(bad)
    assert rc == _lib.ERR_UNSUPPORTED and "not covered" in msg, (rc, msg)
    bad = copy.deepcopy(rn)
    bad.layers[3].flags = 64
    rc, msg = validate(bad)
    assert rc == _lib.ERR_INVALID and "unknown flags" in msg, (rc, msg)

    ag = lowering.lower_graph(graphdef.read_graph(MODEL_PB), "input_1:0", {0: "global_pooling/Mean:0", 1: "age_pred/Softmax:0",
                                                                           2: "gender_pred/Sigmoid:0"}, (96, 96))
    hi = [i for i, x in enumerate(ag.layers) if x.flags & lowering.OPF_HEADS]
    assert len(hi) == 1 and validate(ag)[0] == 0
    bad = copy.deepcopy(ag)
    bad.layers[hi[0]].flags = 0
    bad.layers[hi[0] + 1].flags = lowering.OPF_HEADS                  # on age_pred: not the head of the group
    rc, msg = validate(bad)
    assert rc == _lib.ERR_INVALID and "HEADS" in msg, (rc, msg)
    # truncated and over-long blobs, and the no-GPU contract of the new entry point
    blob = ag.serialize()
    buf = ctypes.create_string_buffer(blob, len(blob))
    assert L.hsefr_plan_validate(ctypes.cast(buf, ctypes.c_void_p), len(blob) - 16) == _lib.ERR_INVALID
    assert L.hsefr_plan_validate(ctypes.cast(buf, ctypes.c_void_p), 40) == _lib.ERR_INVALID
    assert L.hsefr_plan_validate(None, 0) == _lib.ERR_INVALID
    assert L.hsefr_nn1_fallbacks() == 0


def test_round_one_stem_left_the_product_library():
    """csrc/stem_fused.hip (reachable only through lower_graph(stem_fusion="stem")) and csrc/conv3x3_win_bf16.hip are
    development-build sources; a plan that asks for HSEFR_OP_STEM_F16S is refused by the product with a message that says so, and the
    product stays below 4 MB."""
    from conftest import MODEL_PB
    from hse_facerec_tf_amd import _lib, graphdef, lowering
    L = _lib.lib()
    if hasattr(L, "hsefr_debug_set"):
        pytest.skip("a development build is loaded")
    assert not hasattr(L, "hsefr_stem_fused")
    plan = lowering.lower_graph(graphdef.read_graph(MODEL_PB), "input_1:0", {0: "global_pooling/Mean:0"}, (96, 96), stem_fusion="stem")
    assert plan.layers[0].kind == lowering.OP_STEM_F16S
    blob = plan.serialize()
    buf = ctypes.create_string_buffer(blob, len(blob))
    rc = L.hsefr_plan_validate(ctypes.cast(buf, ctypes.c_void_p), len(blob))
    assert rc == _lib.ERR_UNSUPPORTED and "development builds" in _lib.last_error()
    assert os.path.getsize(_lib.LIB_PATH) < 4_000_000
