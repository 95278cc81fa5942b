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


def init_process_group(backend, local_rank=None):
    """Rendezvous from the torchrun environment.  For RCCL (backend 'nccl') pass this process's
    ``local_rank``: the communicator is then bound to that GPU up front instead of being guessed
    at the first collective."""
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        kwargs = {}
        if backend == 'nccl' and local_rank is not None:
            import torch
            kwargs['device_id'] = torch.device('cuda', int(local_rank))
        dist.init_process_group(backend=backend, **kwargs)
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


def classify_fast5_files_sharded(fast5_files, start_model, start_input_size, end_model,
                                 end_input_size, output_size, args):
    """``deepbinner classify DIR`` across the GPUs of one node: every rank (one process per GPU,
    its model(s) already loaded on its own device) classifies a contiguous shard of the sorted
    file list with the ordinary per-batch loop; per-read calls come back with one all-gather
    (RCCL on GPUs), TSV lines and read ids with one object gather; rank 0 prints exactly what the
    single-process path prints (reference classify.py:106-180).  Returns the same
    ``(classifications, read_id_to_fast5_file)`` on rank 0 and ``({}, {})`` elsewhere."""
    import sys
    import torch
    from . import classify as c
    from .load_fast5s import determine_single_or_multi_fast5s
    from .misc import print_summary_table

    rank, local_rank, world = env_world()
    backend = os.environ.get('DEEPBINNER_DIST_BACKEND', 'nccl')
    if backend == 'nccl':
        torch.cuda.set_device(local_rank)       # object collectives stage through this device
    dist = init_process_group(backend, local_rank)
    if not fast5_files:
        sys.exit('Error: no fast5 files found')
    fast5_files = sorted(fast5_files)           # os.walk order may differ between processes
    if determine_single_or_multi_fast5s(fast5_files) == 'multi':
        sys.exit('Error: deepbinner classify requires one-read-per-file fast5s - convert with '
                 'multi_to_single_fast5 before running')
    a, b = shard_bounds(len(fast5_files), world, rank)
    mine = fast5_files[a:b]

    if rank == 0:
        c.print_classification_progress(0, len(fast5_files), 'fast5s')
        c.print_output_header(args.verbose, start_model is not None, end_model is not None,
                              output_size)
    classifications, id_to_file, lines = {}, {}, []
    for loaded in c.load_in_batches(mine, args):
        read_ids, signals = [], []
        for fast5_file, read_id, signal in loaded:
            if signal is None:
                continue
            id_to_file[read_id] = fast5_file
            read_ids.append(read_id)
            signals.append(signal)
        if getattr(loaded, 'complete', False):      # the loader's packed buffer is these reads
            signals = c.PackedSignals(signals, loaded.samples, loaded.offsets)
        lines += c.classify_read_batch(read_ids, signals, start_model, start_input_size,
                                       end_model, end_input_size, output_size, args,
                                       classifications)
        if rank == 0:   # rank 0's shard is as large as any: its progress stands for the job
            c.print_classification_progress(min(len(classifications) * world, len(fast5_files)),
                                            len(fast5_files), 'fast5s')

    # per-read calls: int32, 0 = 'none' (the collective of SURVEY.md section 8e)
    device = torch.device('cuda', local_rank) if backend == 'nccl' else torch.device('cpu')
    order = list(classifications)
    local = torch.tensor([0 if classifications[r] == 'none' else int(classifications[r])
                          for r in order], dtype=torch.int32, device=device)
    counts = [None] * world
    dist.all_gather_object(counts, len(order))
    longest = max(max(counts), 1)
    padded = torch.zeros(longest, dtype=torch.int32, device=device)
    padded[:local.numel()] = local
    gathered = torch.empty(world * longest, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(gathered, padded)
    payload = [None] * world if rank == 0 else None
    dist.gather_object((order, lines, id_to_file), payload, dst=0)
    result = ({}, {})
    if rank == 0:
        all_calls = gathered.cpu().numpy().reshape(world, longest)
        merged, merged_files = {}, {}
        for r, (ids, tsv, files) in enumerate(payload):
            for line in tsv:
                print(line)
            for i, read_id in enumerate(ids):
                merged[read_id] = 'none' if all_calls[r, i] == 0 else str(int(all_calls[r, i]))
            merged_files.update(files)
        c.print_classification_progress(len(merged), len(fast5_files), 'fast5s')
        print('', file=sys.stderr)
        print_summary_table(merged)
        result = (merged, merged_files)
    dist.barrier()
    return result
