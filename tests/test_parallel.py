"""N > 1 path on CPU: world_size-2 gloo processes shard reads, process them independently (no collective on
the data path) and gather in input order; the timing MAX-reduce used by bench.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bonito_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class FakeRead:
    def __init__(self, i, n):
        self.read_id, self.signal = "read_%d" % i, torch.arange(n, dtype=torch.float32).numpy() + i


def fake_basecall(model, reads, scale=1):
    """Stands in for bonito_amd.crf.basecall on CPU: same (read, dict) protocol, deterministic payload."""
    for read in reads:
        yield read, {"sequence": "ACGT"[int(read.signal[0]) % 4] * (len(read.signal) // 100), "n": int(len(read.signal)) * scale}


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = parallel.init("gloo")
    assert (r, w) == (rank, world)
    reads = [FakeRead(i, 300 + 100 * (i % 5)) for i in range(11)]
    got = parallel.basecall_sharded(fake_basecall, None, reads, scale=2)
    slow = parallel.max_over_ranks(1.0 + rank)
    dist.barrier()
    q.put((rank, got, slow))
    dist.destroy_process_group()


def test_two_process_shard_and_ordered_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict()
    for _ in range(world):
        rank, got, slow = q.get(timeout=120)
        outs[rank] = (got, slow)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[1][0] is None                      # only rank 0 holds the result
    got = outs[0][0]
    want = [(r.read_id, res) for r, res in fake_basecall(None, [FakeRead(i, 300 + 100 * (i % 5)) for i in range(11)], scale=2)]
    assert got == want                              # identical to the single-process run, in input order
    assert outs[0][1] == outs[1][1] == 2.0          # MAX over ranks


def test_shard_is_a_partition():
    items = list(range(23))
    parts = [list(parallel.shard(items, r, 4)) for r in range(4)]
    flat = sorted(kv for p in parts for kv in p)
    assert flat == [(i, i) for i in items]
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_gather_is_identity():
    assert parallel.gather_in_order([(2, "c"), (0, "a"), (1, "b")], 0, 1) == ["a", "b", "c"]
    assert parallel.max_over_ranks(3.5) == 3.5
