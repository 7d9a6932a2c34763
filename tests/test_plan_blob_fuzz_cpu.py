"""CPU suite: the host side of libhsefr under AddressSanitizer + UndefinedBehaviorSanitizer.

hsefr_engine_create parses a caller-supplied plan blob (offsets, sizes, buffer ids, kinds: everything the reference hands to
tf.import_graph_def at facerec_test.py:41-48 arrives here as bytes).  csrc/build.sh with HSEFR_ASAN=1 compiles the HOST half of the
product's sources under both sanitizers (flags: csrc/build_asan_flags.sh; no device code is compiled and none runs: GPU sanitizers are not available on
this pool) into csrc/build_asan/libhsefr_asan.so + the driver csrc/fuzz_plan.cc.  The driver pushes >= 12 000 truncated, bit-flipped and
field-mutated copies of six real plans through hsefr_plan_validate and hsefr_engine_create: every one must come back as HSEFR_OK or as a
negative hsefr_status with a message; a crash or a sanitizer report aborts the driver and fails this test."""
import os
import subprocess

import pytest

from conftest import MODEL_PB, ROOT

CSRC = os.path.join(ROOT, "hse_facerec_tf_amd", "csrc")


@pytest.fixture(scope="module")
def asan_build():
    r = subprocess.run(["bash", os.path.join(CSRC, "build.sh")], env=dict(os.environ, HSEFR_ASAN="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    exe = os.path.join(CSRC, "build_asan", "fuzz_plan")
    assert os.path.exists(exe) and os.path.exists(os.path.join(CSRC, "build_asan", "libhsefr_asan.so"))
    return exe


@pytest.fixture(scope="module")
def seed_plans(tmp_path_factory):
    from hse_facerec_tf_amd import graphdef, lowering, resnet50
    d = tmp_path_factory.mktemp("plans")
    g = graphdef.read_graph(MODEL_PB)
    outs = {0: "global_pooling/Mean:0", 1: "age_pred/Softmax:0", 2: "gender_pred/Sigmoid:0"}
    w = resnet50.synthetic_weights(1)
    plans = {
        "mobilenet192": lowering.lower_graph(g, "input_1:0", {0: outs[0]}, (192, 192), input_bound=256.0, u8_mean_bgr=(103.939, 116.779, 123.68)),
        "agegender224": lowering.lower_graph(g, "input_1:0", outs, (224, 224), input_bound=256.0),          # epilogue GEMMs + the fused heads
        "small_batch": lowering.lower_graph(g, "input_1:0", outs, (100, 100), presplit="none"),              # stem3 route, plain GEMMs
        "strict_f32": lowering.lower_graph(g, "input_1:0", outs, (96, 96), pw_math="f32", stem_fusion="none", block_fusion="none"),
        "resnet50_bf16": resnet50.build_plan(w, (224, 224), "caffe"),                                         # projected shortcuts + pairs
        "resnet50_f32": resnet50.build_plan(w, (64, 64), "valid", dtype="f32"),
    }
    paths = []
    for name, p in plans.items():
        path = str(d / (name + ".plan"))
        with open(path, "wb") as fh:
            fh.write(p.serialize())
        paths.append(path)
    return paths


def test_mutated_plan_blobs_never_crash_or_trip_a_sanitizer(asan_build, seed_plans):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([asan_build, "2200"] + seed_plans, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    tail = r.stdout.strip().splitlines()[-1]
    assert tail.startswith("fuzz_plan:") and tail.endswith("0 without a message"), tail
    valid, refused = int(tail.split()[1]), int(tail.split()[5])
    assert valid + refused == 2200 * len(seed_plans) >= 12000 and refused > valid > 0, tail


def test_the_sanitizer_build_is_live(asan_build, tmp_path):
    """The driver must be able to FAIL: a plan file too short to be one is refused by the driver itself (exit 2), and the library under
    test really is the instrumented one (its dynamic symbols carry the ASan runtime's hooks)."""
    bad = tmp_path / "short.plan"
    bad.write_bytes(b"\0" * 10)
    r = subprocess.run([asan_build, "10", str(bad)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2
    out = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(CSRC, "build_asan", "libhsefr_asan.so")], capture_output=True, text=True).stdout
    assert "__asan_report_load" in out and "__ubsan_handle" in out
