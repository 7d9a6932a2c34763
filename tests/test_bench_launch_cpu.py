"""CPU suite, part 6: `python bench.py --gpus N` typed as is starts its own N ranks -- checked
with the GPU-free --dry-run leg (process group on gloo, the product's all-gather), and the pre-launched form the
driver uses still works.  Plus the LFW-shaped label vector of config 5."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _last_json(stdout: str):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout                       # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_self_launches_its_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line == {"dry_run": True, "n_gpus": 2, "gathered_first_column": [0.0, 0.0, 1.0, 1.0], "self_launched": True}


def test_a_failing_rank_fails_the_self_launched_job_with_its_stderr():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSEFR_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "rank 1: injected failure" in r.stderr and "the launched job failed" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]          # no JSON line from a failed job


def test_bench_prelaunched_by_torchrun_does_not_relaunch():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29731", BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["self_launched"] is False


def test_bench_without_gpu_fails_loudly_not_silently():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, BENCH, "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


def test_lfw_like_labels_have_lfw_totals():
    from hse_facerec_tf_amd import gallery
    sizes = gallery.lfw_like_class_sizes()
    assert sizes.sum() == 9164 and len(sizes) == 1680 and sizes.min() == 2 and sizes.max() >= 530
    assert np.all(np.diff(sizes) <= 0)                                  # long tail, sorted
    y = gallery.lfw_like_labels()
    assert np.array_equal(np.bincount(y), sizes) and np.all(np.diff(y) >= 0)      # directory-walk order
    for n, c in ((100, 10), (20, 10), (7, 3)):
        s = gallery.lfw_like_class_sizes(n, c)
        assert s.sum() == n and len(s) == c and s.min() >= 2
