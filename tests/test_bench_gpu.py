"""bench.py on the GPU box: the JSON contract, config 5 end to end, and the N > 1 path (self-launched ranks, the
product's all-gather, per-rank figures) -- with two ranks SHARING the one GPU of the test box over gloo, since RCCL
wants one GPU per rank; the collective call sites and the sharding are the ones the 8-GPU run uses."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def _check_config5(c5, n, world):
    assert "error" not in c5, c5
    assert c5["n_gpus"] == world and c5["shard_rows"] == -(-n // world)
    assert c5["gathered_shard_equals_local"] is True
    assert c5["picks_not_nearest_within_1e-6"] == 0
    assert abs(c5["accuracy"] - c5["accuracy_fp64_bruteforce"]) <= (c5["nn_index_mismatches_vs_fp64"] + 0.5) / (n // 2)
    if c5["accuracy_sklearn"] is not None:
        assert abs(c5["accuracy"] - c5["accuracy_sklearn"]) <= (c5["nn_index_mismatches_vs_fp64"] + 0.5) / (n // 2)
    assert len(c5["extract_ms_per_rank"]) == world and all(v > 0 for v in c5["extract_ms_per_rank"])
    for k in ("allgather_ms", "normalize_ms", "select_ms", "nn1_ms", "total_ms", "extract_faces_per_s"):
        assert c5[k] >= 0


def test_bench_line_contract_and_small_config5():
    line = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--pipeline-files", "300",
                 "--config5-images", "700", "--config5-classes", "120"])
    pl = line["pipeline"]
    assert "error" not in pl and pl["h2d_inclusive"]["value"] > 0 and pl["file_inclusive"]["value"] > 0
    assert pl["file_inclusive"]["value"] <= pl["h2d_inclusive"]["value"] * 1.05
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "config5"):
        assert k in line
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["unit"] == "faces/s" and line["scaling"] == "weak"
    assert line["value"] == pytest.approx(256 * 1e3 / line["ms_per_step"], rel=1e-3)
    rf = line["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3)
    assert "workload" in line["config"] and "model" not in line["config"]
    dw = line["roofline_depthwise"]
    assert "STANDALONE" in dw["covers"] and len(dw["inside_fused_kernels"]) >= 1
    _check_config5(line["config5"], 700, 1)
    assert abs(line["config5"]["unaccounted_ms"]) < 0.25 * line["config5"]["total_ms"]
    su = line["sustained"]
    assert su["seconds"] >= 2.0 and su["steps"] % 100 == 0 and su["window_faces_per_s_min"] <= su["value"] * 1.02 <= su["window_faces_per_s_max"] * 1.04
    assert line["config"]["sustained_faces_per_s"] == su["value"] and line["config"]["config5_total_ms"] == line["config5"]["total_ms"]
    fi = pl["file_inclusive"]
    assert len(fi["host_decode_runs_faces_per_s"]) == 2 and fi["host_decode_faces_per_s"] == max(fi["host_decode_runs_faces_per_s"])
    lat = line["latency_batch1"]
    assert "error" not in lat and lat["age_gender_fun"]["median_ms"] > 0 and lat["extract_features"]["median_ms"] > 0
    assert lat["reference_published"]["age_gender_fun_ms"] == 4.97 and lat["mtcnn_process_image"]["faces"] == 4


def test_bench_two_self_launched_ranks_share_the_gpu_over_gloo():
    line = _run(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--no-op-events",
                 "--config5-images", "701", "--config5-classes", "120"])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 512
    assert len(line["per_rank_faces_per_s"]) == 2 and line["allgather_ms"] > 0
    cfg = line["config"]          # two ranks over gloo on ONE GPU: the line says so itself
    assert cfg["rccl_ranks"] == 2 and cfg["rccl_version"] is None and [r["rank"] for r in cfg["ranks"]] == [0, 1] and cfg["distinct_gpus"] == 1
    assert line["value"] == pytest.approx(2 * 256 * 1e3 / line["ms_per_step"], rel=1e-3)
    c5 = line["config5"]
    _check_config5(c5, 701, 2)
    assert c5["pad_rows"] == 1 and c5["allgather_bytes_per_rank"] == 351 * 1024 * 4
    assert c5["gathered_rows"] == 701 and c5["gathered_rows_differing_from_single_rank_extraction"] == 0


def test_bench_eight_ranks_lfw_shards_with_pad_rows_equal_the_single_rank_extraction():
    """BASELINE configs[4] at its own size and world size -- 9164 photos over EIGHT ranks (S = 1146, the last
    shard 1142 rows + 4 pad rows) -- with the ranks sharing this box's one GPU over gloo: every call site of the 8-GPU job
    (torchrun env, shard ranges, the ONE all-gather, identification on every rank) and the gathered matrix row for row against
    rank 0's single-rank extraction of all 9164 photos."""
    line = _run(["--gpus", "8", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-op-events", "--no-sustained"], timeout=1500)
    assert line["n_gpus"] == 8 and len(line["per_rank_faces_per_s"]) == 8
    c5 = line["config5"]
    _check_config5(c5, 9164, 8)
    assert c5["shard_rows"] == 1146 and c5["pad_rows"] == 4 and c5["allgather_bytes_per_rank"] == 1146 * 1024 * 4
    assert c5["gathered_rows"] == 9164 and c5["gathered_rows_differing_from_single_rank_extraction"] == 0
    assert c5["gathered_shard_equals_local"] is True and c5["num_classes"] == 1680


def test_side_configs_carry_a_per_kernel_view():
    """other_configs[ResNet-50 bf16] and other_configs[age / gender] carry `kernels` (one row per kernel instantiation,
    named by the launchers' own routing) and the dominant kernel as a roofline object of the headline's shape; the strict-fp32 ResNet
    carries `traffic` once a profile of it is committed."""
    line = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pipeline", "--no-latency", "--no-config5", "--no-sustained", "--no-op-events"])
    oc = {o["config"].split(":")[0]: o for o in line["other_configs"]}
    for key, fam in (("BASELINE configs[2]", "conv"), ("BASELINE configs[3]", "pwconv_ps_kernel")):
        o = oc[key]
        assert "error" not in o, o
        ks, rf = o["kernels"], o["roofline_dominant_kernel"]
        assert len(ks) >= 5 and ks == sorted(ks, key=lambda k: -k["ms_per_step"]) and rf["kernel"] == ks[0]["kernel"]
        assert any(k["kernel"].startswith(fam) for k in ks)
        for k in ks:
            assert k["bound"] in ("hbm", "mfma") and k["frac"] == pytest.approx(k["achieved"] / k["peak"], rel=2e-3, abs=2e-4) and k["launches_per_step"] >= 1
            for f in ("avg_launch_us", "algorithmic_bytes_per_launch", "traffic", "traffic_stale", "mfma_util_pmc", "floors_us"):
                assert f in k
        assert abs(sum(k["ms_per_step"] for k in ks) - o["ms_per_step"]) < 0.35 * o["ms_per_step"]          # (event packets cost a few percent)
        for f in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches_per_step"):
            assert f in rf
    assert any("conv1x1_pair_bf16_kernel" in k["kernel"] for k in oc["BASELINE configs[2]"]["kernels"])
    assert any(k["kernel"] == "heads_kernel" for k in oc["BASELINE configs[3]"]["kernels"])
    f32 = next(o for o in line["other_configs"] if o["config"].startswith("BASELINE configs[2] in the fp32-grade mode"))
    assert "traffic" in f32["roofline"] and "traffic_source" in f32["roofline"]
