"""CPU suite, part 5: the multi-GPU path (contiguous sharding + ONE all-gather) on world_size-2
gloo.  The extractor is injected (a CPU stand-in), the collective and the sharding are the
product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hse_facerec_tf_amd import gallery


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_extract(items):
    # embedding = deterministic function of the item id, so order and padding are checkable
    ids = torch.tensor(items, dtype=torch.float32).reshape(-1, 1)
    return torch.cat([ids, ids * 2 + 1, torch.sin(ids), torch.ones_like(ids)], dim=1)


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items = list(range(n))
        full = gallery.extract_sharded(_fake_extract, items, 4, torch.device("cpu"), batch=3)
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [11, 8, 1])
def test_sharded_extract_all_gather_world2(n):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = _fake_extract(list(range(n))).numpy()
    for r in (0, 1):
        assert res[r].shape == (n, 4)
        assert np.array_equal(res[r], want)         # file order preserved, pad rows trimmed, same on every rank


def test_shard_ranges_cover_lfw_exactly():
    n, p = 9164, 8                                   # S = 1146, 4 pad rows
    assert gallery.shard_size(n, p) == 1146
    spans = [gallery.shard_range(n, r, p) for r in range(p)]
    assert spans[0] == (0, 1146) and spans[-1] == (8022, 9164)
    assert sum(hi - lo for lo, hi in spans) == n
    assert all(spans[i][1] == spans[i + 1][0] for i in range(p - 1))
    assert gallery.shard_range(3, 7, 8) == (3, 3)    # more ranks than items: empty tail shards
    assert gallery.shard_size(0, 8) == 0


def test_single_process_path_needs_no_process_group():
    full = gallery.extract_sharded(_fake_extract, list(range(5)), 4, torch.device("cpu"), batch=2)
    assert np.array_equal(full.numpy(), _fake_extract(list(range(5))).numpy())


def test_on_issued_hook_runs_once_after_the_last_batch():
    calls = []

    def extract(ids):
        calls.append(("batch", list(ids)))
        return _fake_extract(ids)
    gallery.extract_sharded(extract, list(range(5)), 4, torch.device("cpu"), batch=2, on_issued=lambda: calls.append(("issued",)))
    assert calls == [("batch", [0, 1]), ("batch", [2, 3]), ("batch", [4]), ("issued",)]
