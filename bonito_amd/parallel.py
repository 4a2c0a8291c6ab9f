"""
Multi-GPU execution of the hot path: replicas, shard-by-read, no data-path collective.

Read chunks are independent (/root/reference bonito/crf/basecall.py:70-72) and the models are small
(0.4-70 M parameters), so every GPU runs its own engine replica on its own process
(``torchrun --nproc-per-node N``; ``torch.distributed`` backend "nccl" = RCCL on ROCm, "gloo" on CPU).
The only communication is (a) the final gather of basecalls to rank 0, which restores the input order the
reference's writer expects, and (b) barriers / a MAX-reduce for timing. Both are off the per-batch path.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard(items, rank, world):
    """Round-robin shard of an iterable: yields (global_index, item) for this rank. Round-robin keeps
    the shards balanced when reads arrive sorted by length and needs no knowledge of the total count."""
    for i, item in enumerate(items):
        if i % world == rank:
            yield i, item


def gather_in_order(indexed_results, rank=None, world=None, dst=0):
    """`indexed_results`: iterable of (global_index, payload) produced by this rank. Returns the payloads
    of ALL ranks in global order on rank `dst` (None elsewhere). Payloads must be picklable."""
    if rank is None or world is None:
        rank, world, _ = env_rank_world()
    mine = list(indexed_results)
    if world == 1:
        return [p for _, p in sorted(mine, key=lambda kv: kv[0])]
    gathered = [None] * world if rank == dst else None
    dist.gather_object(mine, gathered, dst=dst)
    if rank != dst:
        return None
    merged = sorted((kv for part in gathered for kv in part), key=lambda kv: kv[0])
    idx = [k for k, _ in merged]
    assert idx == list(range(len(idx))), "shards do not cover the input exactly once"
    return [p for _, p in merged]


def max_over_ranks(value, device=None):
    """MAX-reduce a python float over all ranks (wall-clock of the slowest replica)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def basecall_sharded(basecall_fn, model, reads, **kwargs):
    """Run `basecall_fn(model, reads_of_this_rank, **kwargs)` on every rank and return all
    (read_id, result) pairs in input order on rank 0."""
    rank, world, _ = env_rank_world()
    mine = list(shard(reads, rank, world))
    index_of = {id(read): i for i, read in mine}
    out = ((index_of[id(read)], (getattr(read, "read_id", None), res))
           for read, res in basecall_fn(model, (r for _, r in mine), **kwargs))
    return gather_in_order(out, rank, world)
