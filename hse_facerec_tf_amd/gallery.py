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


def extract_sharded(extract_fn: Callable, items: Sequence, dim: int, device, group=None, batch: int = 256):
    """extract_fn(list_of_items) -> float32 tensor [len, dim] on ``device``.  Returns the full
    [N, dim] tensor (on ``device``) on every rank.  Works un-initialised (world = 1) too."""
    import torch
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    n = len(items)
    s = shard_size(n, world)
    lo, hi = shard_range(n, rank, world)
    local = torch.zeros((s, dim), dtype=torch.float32, device=device)      # tail rows stay zero (pad)
    for i in range(lo, hi, batch):
        j = min(i + batch, hi)
        out = extract_fn(items[i:j])
        if tuple(out.shape) != (j - i, dim):
            raise ValueError("extract_fn returned %r for %d items of dim %d" % (tuple(out.shape), j - i, dim))
        local[i - lo:j - lo] = out
    if world == 1:
        return local[:n]
    full = torch.empty((world * s, dim), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(full, local, group=group)
    return full[:n]


def gather_labels(labels: np.ndarray) -> np.ndarray:
    """Labels come from the directory names of the (identical, sorted) file list on every rank
    (facerec_test.py:386-389) -- nothing to exchange."""
    return np.asarray(labels)
