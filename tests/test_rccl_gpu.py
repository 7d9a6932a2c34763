"""The N > 1 path's collective on the hardware a test box has: ONE rank on the `nccl` backend (RCCL).  A world-1
communicator still runs ncclCommInitRank, ncclAllGather and the barrier kernels, so the first 8-GPU run is not the
first execution of `dist.init_process_group("nccl", device_id=...)`, `gallery.all_gather_rows` on CUDA tensors or
bench.py's grouped legs.  The collective sits between facerec_test.py:394 and :401."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert not dist.is_initialized()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)       # exactly bench.py's call
    try:
        yield dev
    finally:
        dist.destroy_process_group()


def test_all_gather_rows_on_cuda_tensors_over_rccl(nccl_world1):
    import torch
    import torch.distributed as dist
    from hse_facerec_tf_amd import gallery
    dev = nccl_world1
    assert dist.get_backend() == "nccl"
    local = torch.arange(1146 * 1024, dtype=torch.float32, device=dev).reshape(1146, 1024)      # LFW's shard: [S, D]
    full = gallery.all_gather_rows(local)
    torch.cuda.synchronize()
    assert full.is_cuda and full.shape == (1146, 1024) and full.data_ptr() != local.data_ptr()
    assert torch.equal(full, local)
    maps = open("/proc/self/maps").read()
    assert "librccl" in maps or "libnccl" in maps, "the nccl backend did not load RCCL"


def test_extract_sharded_through_the_engine_and_rccl(nccl_world1):
    """bench.py's config-5 leg in small: device photos -> extract_images -> extract_sharded -> ONE all-gather (RCCL) ->
    1-NN; gathered rows must be the local rows bit for bit, the timings must show the collective ran."""
    import torch
    from hse_facerec_tf_amd import gallery, identification
    from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB, TensorFlowInference
    dev = nccl_world1
    tfi = TensorFlowInference(AGE_GENDER_PB, "input_1:0", "global_pooling/Mean:0", input_size=(96, 96), max_batch=32, device=0)
    n = 75
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    photos = torch.randint(0, 256, (n, 120, 120, 3), dtype=torch.uint8, device=dev, generator=g)

    def extract(ids):
        return tfi.extract_images(photos[ids[0]:ids[-1] + 1])
    timings = {}
    X = gallery.extract_sharded(extract, list(range(n)), tfi.feature_dim, dev, batch=32, timings=timings)
    assert X.shape == (n, 1024) and timings["allgather_s"] > 0 and timings["allgather_bytes_per_rank"] == n * 1024 * 4
    again = torch.cat([extract(list(range(i, min(i + 32, n)))) for i in range(0, n, 32)])
    assert torch.equal(X, again)
    y = np.arange(n) // 3
    res = identification.one_nn_identification(X, y)
    assert 0.0 <= res["accuracy"] <= 1.0 and len(res["test"]) + len(res["train"]) == n
    tfi.close_session()


def _run_bench(extra, launcher):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), BENCH] + extra
    else:
        cmd = [sys.executable, BENCH] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.parametrize("launcher", [False, True], ids=["plain", "torchrun-prelaunched"])
def test_bench_force_group_runs_every_collective_on_rccl(launcher):
    """`bench.py --gpus 1 --force-group`: process group on nccl, barrier-bracketed timed region, the per-rank all-gather of
    timings, the embeddings all-gather, and config 5 through extract_sharded -- the N > 1 code path with N = 1."""
    line = _run_bench(["--gpus", "1", "--force-group", "--steps", "3", "--warmup", "1", "--no-op-events", "--no-cpu-baseline",
                       "--no-other-configs", "--no-pipeline", "--no-latency", "--config5-images", "700", "--config5-classes", "120"], launcher)
    assert line["n_gpus"] == 1 and line["config"]["backend"].startswith("RCCL") and line["config"]["process_group"] is True
    assert line["allgather_ms"] is not None and line["allgather_ms"] > 0
    # the line certifies its own ranks: communicator size, RCCL version, every rank's device through the group itself
    cfg = line["config"]
    assert cfg["rccl_ranks"] == 1 and cfg["rccl_version"] and cfg["rccl_version"][0].isdigit() and cfg["distinct_gpus"] == 1
    assert len(cfg["ranks"]) == 1 and cfg["ranks"][0]["rank"] == 0 and cfg["ranks"][0]["device_name"] and ":" in cfg["ranks"][0]["pci_bus_id"]
    assert cfg["allgather_gbs"] == pytest.approx(256 * 1024 * 4 / (line["allgather_ms"] * 1e-3) / 1e9, rel=0.02)
    c5 = line["config5"]
    assert "error" not in c5, c5
    assert c5["gathered_shard_equals_local"] is True and c5["allgather_ms"] > 0
    assert c5["allgather_bytes_per_rank"] == 700 * 1024 * 4 and c5["picks_not_nearest_within_1e-6"] == 0
