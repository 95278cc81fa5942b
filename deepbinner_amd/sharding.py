"""
Multi-GPU layer of the classify path: reads are independent, so they shard across ranks with no
data-path collective; the only exchange is a gather of per-read barcode calls
(SURVEY.md §8e).  One process per GPU, launched by ``torch.distributed.run``; the process group is
RCCL (backend "nccl") on GPUs and gloo in CPU tests.  torch is imported lazily — the single-GPU
product path never needs it.

The reference has no counterpart (single process, single device, ``classify.py:416-423``).
"""

import os


def shard_bounds(n_items, world_size, rank):
    """Contiguous block of ``n_items`` owned by ``rank``: sizes differ by at most one, earlier
    ranks take the remainder, concatenating the blocks in rank order restores the input order."""
    base, extra = divmod(int(n_items), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)),
            int(os.environ.get('WORLD_SIZE', 1)))


def init_process_group(backend):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group(backend=backend)
    return dist


def gather_calls(local_calls, n_total, world_size, rank):
    """All-gather the per-read int32 calls of every rank's shard into one array of ``n_total``
    calls in read order.  ``local_calls`` is a torch int32 tensor (GPU for RCCL, CPU for gloo)
    holding this rank's shard.  Shards are padded to equal length for the collective."""
    import torch
    import torch.distributed as dist
    longest = -(-n_total // world_size)
    padded = torch.zeros(longest, dtype=torch.int32, device=local_calls.device)
    padded[:local_calls.numel()] = local_calls
    out = torch.empty(world_size * longest, dtype=torch.int32, device=local_calls.device)
    dist.all_gather_into_tensor(out, padded)
    pieces = []
    for r in range(world_size):
        a, b = shard_bounds(n_total, world_size, r)
        pieces.append(out[r * longest:r * longest + (b - a)])
    return torch.cat(pieces)
