"""Gallery extraction sharded over the GPUs of one node (new relative to the
single-process reference loop facerec_test.py:377-399).

One process per GPU (torchrun-style env).  The file list is fixed in a deterministic order,
rank r takes the contiguous slice [r*S, min((r+1)*S, N)), S = ceil(N/P); every rank pads its
embeddings to [S, D] and ONE all-gather (RCCL over xGMI with backend "nccl"; gloo in the CPU
tests) gives every rank the [N, D] matrix in file order.  There is no other data-path
collective: the extract phase is embarrassingly parallel, labels and the split are computed
identically on every rank from the same deterministic inputs.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import numpy as np


def shard_size(n: int, world: int) -> int:
    return (n + world - 1) // world if n > 0 else 0


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    s = shard_size(n, world)
    lo = min(rank * s, n)
    return lo, min(lo + s, n)


def _sync(device) -> float:
    import time
    import torch
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    return time.perf_counter()


def extract_sharded(extract_fn: Callable, items: Sequence, dim: int, device, group=None, batch: int = 256,
                    timings: Optional[dict] = None, on_issued: Optional[Callable] = None):
    """extract_fn(list_of_items) -> float32 tensor [len, dim] on ``device``.  Returns the full
    [N, dim] tensor (on ``device``) on every rank.  Works un-initialised (world = 1) too.
    ``timings`` (optional dict) receives wall seconds of the two phases, device-synchronised:
    'extract_s' (this rank's shard) and 'allgather_s' (the one collective).
    ``on_issued`` (optional) is called once, after the last batch of this rank's shard has been ENQUEUED and before anything
    waits for the device: the place to start host work that needs no features (identification.start_split)."""
    import torch
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    n = len(items)
    s = shard_size(n, world)
    lo, hi = shard_range(n, rank, world)
    local = torch.zeros((s, dim), dtype=torch.float32, device=device)      # tail rows stay zero (pad)
    t0 = _sync(device) if timings is not None else 0.0
    for i in range(lo, hi, batch):
        j = min(i + batch, hi)
        out = extract_fn(items[i:j])
        if tuple(out.shape) != (j - i, dim):
            raise ValueError("extract_fn returned %r for %d items of dim %d" % (tuple(out.shape), j - i, dim))
        local[i - lo:j - lo] = out
    if on_issued is not None:
        on_issued()
    if timings is not None:
        t1 = _sync(device)
        timings["extract_s"] = t1 - t0
        timings["shard"] = (lo, hi)
    if not distributed:                       # no process group: nothing to exchange
        if timings is not None:
            timings["allgather_s"] = 0.0
        return local[:n]
    full = all_gather_rows(local, group)      # also with ONE rank in the group: the collective still runs (a world-1 communicator)
    if timings is not None:
        timings["allgather_s"] = _sync(device) - t1
        timings["allgather_bytes_per_rank"] = int(local.numel() * 4)
    return full[:n]


def all_gather_rows(local, group=None):
    """THE collective of the path: every rank contributes [S, D] and receives [P*S, D] in rank order
    (ncclAllGather on RCCL over xGMI with backend "nccl").  On a gloo group (CPU tests, or several ranks
    sharing one GPU in the launcher test) device tensors are staged through the host."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    full = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(full.shape, dtype=local.dtype)
        dist.all_gather_into_tensor(host, local.cpu().contiguous(), group=group)
        full.copy_(host)
    else:
        dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    return full


def lfw_like_class_sizes(n_images: int = 9164, n_classes: int = 1680) -> np.ndarray:
    """A deterministic long-tailed class-size histogram with LFW's totals after the >1-image filter of
    facerec_test.py:407-412 (9164 photos of 1680 persons, README.md:14): Zipf-like, largest class 530
    (LFW's own maximum), every class >= 2.  The real per-person counts need the dataset; only the totals
    and the shape of the tail matter for the synthetic config-5 workload."""
    r = np.arange(1, n_classes + 1, dtype=np.float64)
    top = min(530, max(2, n_images - 2 * (n_classes - 1)))
    lo, hi = 0.0, 8.0
    for _ in range(80):
        p = 0.5 * (lo + hi)
        if np.maximum(2, np.floor(top / r ** p)).sum() > n_images:
            lo = p
        else:
            hi = p
    sizes = np.maximum(2, np.floor(top / r ** hi)).astype(np.int64)
    sizes[0] += n_images - int(sizes.sum())               # whatever the floor left over goes to the largest class
    if sizes.min() < 2 or sizes.sum() != n_images:
        raise ValueError("no histogram with %d images in %d classes of >= 2" % (n_images, n_classes))
    return sizes


def lfw_like_labels(n_images: int = 9164, n_classes: int = 1680) -> np.ndarray:
    """Labels in directory-walk order (get_files: images of one person are consecutive, persons sorted)."""
    return np.repeat(np.arange(n_classes), lfw_like_class_sizes(n_images, n_classes))
