"""
Multi-GPU execution of the hot path: replicas, shard-by-read, no data-path collective.

Read chunks are independent (/root/reference bonito/crf/basecall.py:70-72) and the models are small
(0.4-70 M parameters), so every GPU runs its own engine replica in its own process
(``python -m bonito_amd basecaller --devices 0-7`` spawns them; ``torchrun --nproc-per-node N`` works too;
``torch.distributed`` backend "nccl" = RCCL on ROCm for the bench barrier, "gloo" for host objects).

* Reads are sharded round-robin at the RECORD level (`Reader.get_reads(rank=, world=)`): a rank only loads, normalises and
  chunks its own reads.
* Every rank formats its own records (`io.format_record`: the FASTQ / SAM text, the summary row, the log entry), so the host
  work of writing scales with the ranks.
* Rank 0 merges the ranks' record streams back into input order and is the only writer (`ordered_records`): record i comes
  from rank i % world, every rank emits its records in order, so the merge pulls from per-rank FIFO streams in turn -- a
  bounded reorder window (a rank produces into a queue of `window` messages that a sender thread drains; the pipeline only
  waits once that window is full), nothing is held until the end.
  The reference gets its ordering from a single process (bonito/io.py:400-469 consumes one iterator); this is the same
  contract across processes.
* A rank that dies does not take the run down (round 5): rank 0 keeps what it received from it and produces the rest of that rank's
  shard itself (`ordered_records(..., rescue=)`); the launcher lets the survivors finish.
* The only collectives are barriers / a MAX-reduce for timing (bench.py). Nothing per batch.
"""
import io
import os
import pickle

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


RENDEZVOUS_TIMEOUT_S = 180.0        # store rendezvous + the collectives that build the groups; BONITO_AMD_RENDEZVOUS_TIMEOUT overrides
STREAM_TIMEOUT_S = 7 * 24 * 3600.0    # point-to-point record streams: a rank legitimately waits as long as the run lasts


def init(backend=None, timeout=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process). The default group gets a SHORT timeout
    (advisor, round 5: a worker that dies before or during the rendezvous - out of memory, a bad device, an import error - used to
    leave the others in `init_process_group` for gloo's default of 30 minutes); the record streams run on `host_group()`, whose
    point-to-point operations may wait as long as the run lasts."""
    import datetime
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if timeout is None:
            timeout = float(os.environ.get("BONITO_AMD_RENDEZVOUS_TIMEOUT", RENDEZVOUS_TIMEOUT_S))
        td = datetime.timedelta(seconds=timeout)
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local), timeout=td)
        else:
            dist.init_process_group(backend, timeout=td)
    return rank, world, local


def rendezvous_done():
    """Tell the launcher (cli/basecaller.py `launch`) that every rank has joined: from here on a lost worker is absorbed by rank 0; before,
    the launcher takes the run down. The file named by BONITO_AMD_READY_FILE is created by rank 0 once the host group exists."""
    path = os.environ.get("BONITO_AMD_READY_FILE")
    if path and env_rank_world()[0] == 0:
        with open(path, "w") as fh:
            fh.write("ready\n")


def shard(items, rank, world):
    """Round-robin shard of an iterable: yields (global_index, item) for this rank. Round-robin keeps
    the shards balanced when reads arrive sorted by length and needs no knowledge of the total count."""
    for i, item in enumerate(items):
        if i % world == rank:
            yield i, item


def max_over_ranks(value, device=None):
    """MAX-reduce a python float over all ranks (wall-clock of the slowest replica)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- host-object point-to-point (gloo): length-prefixed pickles ------------------------------------------------
def _send_obj(obj, dst, group):
    buf = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    dist.send(torch.tensor([len(buf)], dtype=torch.int64), dst=dst, group=group)
    if buf:
        dist.send(torch.frombuffer(bytearray(buf), dtype=torch.uint8), dst=dst, group=group)


def _recv_obj(src, group):
    n = torch.zeros(1, dtype=torch.int64)
    dist.recv(n, src=src, group=group)
    data = torch.empty(int(n.item()), dtype=torch.uint8)
    if data.numel():
        dist.recv(data, src=src, group=group)
    return pickle.load(io.BytesIO(data.numpy().tobytes()))


# ---- packed record blocks (round 6): the product's records are (text, summary_row, log) triples of ~40 KB each, tens of thousands per
# second across eight ranks. Pickling them one by one made rank 0's merge thread the bottleneck of the whole node (4 ranks on 8 cores:
# 1.10e9 samples/s without the merge, 0.905e9 through it). A message is now ONE buffer: a small header, four int64 arrays (text / row / id
# lengths, samples) and three blobs; the worker renders the summary row with the csv dialect of io.Writer, rank 0 slices memoryviews out
# of the received buffer and writes them as bytes - no unpickling, no str round trip of the record text.
_PACK_MAGIC = 0x626C6B31          # "blk1"


def render_summary_row(row, _cache={}):
    """bytes of `csv.writer(fh, delimiter="\t").writerow(row)` (what io.Writer writes into summary.tsv)."""
    import csv
    if "w" not in _cache:
        _cache["buf"] = io.StringIO()
        _cache["w"] = csv.writer(_cache["buf"], delimiter="\t")
    buf = _cache["buf"]
    buf.seek(0)
    buf.truncate()
    _cache["w"].writerow(row)
    return buf.getvalue().encode("utf-8")


def pack_records(records, last):
    """list of (text: str | bytes | None, summary_row: list | bytes | None, (read_id, samples)) -> one bytearray."""
    import numpy as np
    n = len(records)
    meta = np.empty((4, n), np.int64)
    texts, rows, ids = [], [], []
    for i, (text, row, log) in enumerate(records):
        if text is None:
            meta[0, i] = -1
        else:
            if isinstance(text, str):
                text = text.encode("utf-8")
            texts.append(text)
            meta[0, i] = len(text)
        if row is None:
            meta[1, i] = -1
        else:
            if not isinstance(row, (bytes, bytearray, memoryview)):
                row = render_summary_row(row)
            rows.append(row)
            meta[1, i] = len(row)
        rid = log[0].encode("utf-8")
        ids.append(rid)
        meta[2, i] = len(rid)
        meta[3, i] = int(log[1])
    head = np.array([_PACK_MAGIC, n, 1 if last else 0, 0], np.int64)
    return bytearray(b"".join([head.tobytes(), meta.tobytes(), *texts, *rows, *ids]))


def unpack_records(buf):
    """The inverse, zero-copy: -> (list of (text memoryview | None, row bytes | None, (read_id, samples)), last)."""
    import numpy as np
    mv = memoryview(buf)
    head = np.frombuffer(mv[:32], np.int64)
    if int(head[0]) != _PACK_MAGIC:
        raise ValueError("not a packed record block")
    n, last = int(head[1]), bool(head[2])
    meta = np.frombuffer(mv[32:32 + 32 * n], np.int64).reshape(4, n)
    tl, rl, il, samples = (meta[k].tolist() for k in range(4))
    pos = 32 + 32 * n
    t_off, r_off = [], []
    for v in tl:
        t_off.append(pos)
        pos += max(v, 0)
    for v in rl:
        r_off.append(pos)
        pos += max(v, 0)
    out = []
    for i in range(n):
        rid = bytes(mv[pos:pos + il[i]]).decode("utf-8")
        pos += il[i]
        out.append((mv[t_off[i]:t_off[i] + tl[i]] if tl[i] >= 0 else None,
                    bytes(mv[r_off[i]:r_off[i] + rl[i]]) if rl[i] >= 0 else None, (rid, samples[i])))
    return out, last


def _send_buf(buf, dst, group):
    dist.send(torch.tensor([len(buf)], dtype=torch.int64), dst=dst, group=group)
    if len(buf):
        dist.send(torch.frombuffer(buf, dtype=torch.uint8), dst=dst, group=group)


def _recv_buf(src, group):
    n = torch.zeros(1, dtype=torch.int64)
    dist.recv(n, src=src, group=group)
    data = torch.empty(int(n.item()), dtype=torch.uint8)
    if data.numel():
        dist.recv(data, src=src, group=group)
    return data.numpy()


_HOST_GROUP = None


def host_group():
    """A gloo group for host objects (the default group may be RCCL, which only moves device tensors), with a timeout that does not
    bound the run: a worker that has finished its shard waits for rank 0's closing message for as long as the slowest rank needs.
    Creating it is a collective of every rank (bounded by the default group's short timeout)."""
    global _HOST_GROUP
    if _HOST_GROUP is None:
        import datetime
        _HOST_GROUP = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=STREAM_TIMEOUT_S))
        rendezvous_done()
    return _HOST_GROUP


def shutdown():
    """Normal exit path of a multi-process run: drop the groups (advisor, round 5: the CLI no longer did)."""
    global _HOST_GROUP
    _HOST_GROUP = None
    if dist.is_available() and dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception:          # a peer that died has left a half-open pair behind: nothing to clean up on its side
            pass


_PEER_GONE = ("closed by peer", "reset by peer", "connection closed", "broken pipe", "socket closed", "connection refused",
              "peer closed", "connection reset")


def peer_is_gone(exc):
    """Is this gloo failure a DEAD peer (connection closed / reset) rather than a live one that is slow, a timeout or a corrupt message?
    Only the former may be rescued: a stalled-but-alive rank keeps sending into its stream, and a duplicate producer for its shard
    would leave it blocked for ever (advisor, round 5)."""
    text = str(exc).lower()
    return isinstance(exc, (RuntimeError, ConnectionError, OSError)) and any(tok in text for tok in _PEER_GONE)


class _Prefetch:
    """Runs an iterator in a thread and hands its items over a bounded queue: the producer (a rank's basecalling pipeline) keeps
    running while the consumer is blocked elsewhere (a send to rank 0, a receive from a slower rank), up to `depth` items ahead;
    beyond that it waits - memory stays bounded. Exceptions of the producer surface in the consumer."""
    _END = object()

    def __init__(self, iterator, depth):
        import queue
        import threading
        self.q = queue.Queue(maxsize=max(1, depth))
        self.exc = None
        self.t = threading.Thread(target=self._run, args=(iterator,), daemon=True)
        self.t.start()

    def _run(self, iterator):
        try:
            for item in iterator:
                self.q.put(item)
        except BaseException as exc:      # handed to the consumer
            self.exc = exc
        self.q.put(self._END)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is self._END:
                if self.exc is not None:
                    raise self.exc
                return
            yield item


_BYE = "__bonito_amd_bye__"


def ordered_records(local_records, rank=None, world=None, batch=64, group=None, window=32, rescue=None, on_rank_lost=None, packed=False):
    """Merge the per-rank record streams into global input order on rank 0.

    `local_records`: this rank's records in ITS order; its k-th record is global record ``rank + k * world`` (what
    `Reader.get_reads(rank=, world=)` / `shard` produce). Records must be picklable.

    Rank 0: returns a generator over ALL records in global order (record i is pulled from rank i % world's stream; the
    stream of a rank is a sequence of messages of up to `batch` records, the last one flagged). Other ranks: the call
    sends this rank's records to rank 0 as they are produced, waits for rank 0's closing message and returns an empty iterator.

    Back-pressure without stalling the GPUs: on every rank the records are produced by a thread of their own into a queue of at
    most `window` messages (`window * batch` records), so a rank whose send is waiting for rank 0 - because an EARLIER record of a
    slower rank is still missing - keeps basecalling until that window is full, and rank 0's own pipeline keeps running while
    its writer waits for another rank. Nothing is ever held beyond the windows (a run with one rank ten times slower than the
    others finishes with the same bytes, tests/test_parallel.py).

    A rank that DIES (round 5; SURVEY 5 "failure detection": re-queue on another replica): records are idempotent and keyed by their
    index, so when rank 0 can no longer receive from rank r it keeps what it has received and takes the rest of r's stream from
    ``rescue(r, k)`` - an iterator over rank r's records from its k-th on, produced by rank 0's own pipeline (the CLI builds it from
    the same reader shard and the same model) - instead of taking the run down. `on_rank_lost(r, k, exc)` is told. Without `rescue` the
    failure propagates as before. The streams end with a closing message from rank 0 to every rank that is still there (no
    collective: a barrier would wait for the dead).

    `packed` (round 6, the CLI's setting): the records are the product's (text, summary_row, (read_id, samples)) triples and travel as
    packed blocks (`pack_records`) instead of pickles; rank 0 then yields (text memoryview | None, row bytes | None, log) for the peers'
    records - io.Writer writes both forms to the same bytes."""
    if rank is None or world is None:
        rank, world, _ = env_rank_world()
    if world == 1:
        return iter(local_records)
    group = group or host_group()

    def messages(records):
        pending = []
        for rec in records:
            pending.append(rec)
            if len(pending) == batch:
                yield pending, False
                pending = []
        yield pending, True

    if rank != 0:
        if packed:
            # the blocks are built in the producer's thread (behind the prefetch queue), the sender only moves bytes
            for blob in _Prefetch((pack_records(recs, last) for recs, last in messages(local_records)), window):
                _send_buf(blob, 0, group)
        else:
            for msg in _Prefetch(messages(local_records), window):
                _send_obj(msg, 0, group)
        bye = _recv_obj(0, group)                 # rank 0 has everything (raises if rank 0 is gone)
        assert bye == _BYE, "unexpected closing message %r" % (bye,)
        return iter(())

    def merged():
        local = iter(_Prefetch(local_records, window * batch))
        bufs = [[] for _ in range(world)]        # records received from rank r and not yet emitted
        done = [False] * world
        got = [0] * world                        # records received from rank r so far
        lost = [None] * world                    # rank r died: the iterator that stands in for the rest of its stream

        def receive(src):
            """next message of rank src -> bufs / done; a dead peer switches the stream over to the rescue iterator"""
            try:
                recs, last = unpack_records(_recv_buf(src, group)) if packed else _recv_obj(src, group)
            except Exception as exc:
                # only a peer whose connection is CLOSED / RESET is dead; a timeout or a garbled message of a live rank is an error
                if rescue is None or not peer_is_gone(exc):
                    raise
                if on_rank_lost is not None:
                    on_rank_lost(src, got[src], exc)

                def lazily(r=src, k=got[src]):          # the stand-in producer (a second model on rank 0's GPU in the CLI) is built on
                    yield from rescue(r, k)             # the first record asked of it - a rank that died BEHIND its last record costs nothing
                lost[src] = lazily()
                return
            bufs[src] = recs
            got[src] += len(recs)
            done[src] = last

        i = 0
        while True:
            src = i % world
            if src == 0:
                try:
                    yield next(local)
                except StopIteration:
                    break
            else:
                while not bufs[src] and not done[src] and lost[src] is None:
                    receive(src)
                if bufs[src]:
                    yield bufs[src].pop(0)
                elif lost[src] is not None:
                    try:
                        yield next(lost[src])
                    except StopIteration:
                        break                    # rank src's shard is exhausted: indices are dense, so nothing follows i
                else:
                    break                        # rank src is exhausted
            i += 1
        for r in range(1, world):                # drain the final (possibly empty) messages so no sender is left blocked
            while not done[r] and lost[r] is None:
                receive(r)
                assert not bufs[r], "rank %d holds records beyond the end of the stream" % r
            # (a lost rank's stand-in is NOT asked for "one more record" here: the indices are dense, so once the loop above has ended
            #  every shard is exhausted by construction, and asking would build the stand-in - a second model on rank 0's GPU in the CLI -
            #  for a rank that died behind its last record only to learn that nothing is left; advisor, round 5)
        assert not any(bufs), "records left over after the merge"
        for r in range(1, world):
            if lost[r] is None:
                try:
                    _send_obj(_BYE, r, group)
                except Exception:                # it died after its last message: nothing left to tell it
                    pass

    return merged()


def format_stream(results, mode, min_qscore=0.0):
    """(read, result) pairs of this rank -> the picklable (text, summary_row, log) triples rank 0 writes."""
    from bonito_amd.io import format_record
    for read, res in results:
        yield format_record(read, res, mode, min_qscore)


def gather_in_order(indexed_results, rank=None, world=None, dst=0):
    """`indexed_results`: iterable of (global_index, payload) produced by this rank. Returns the payloads of ALL
    ranks in global order on rank `dst` (None elsewhere). Small results only (everything is held in memory): the
    product path streams through `ordered_records` instead."""
    if rank is None or world is None:
        rank, world, _ = env_rank_world()
    mine = list(indexed_results)
    if world == 1:
        return [p for _, p in sorted(mine, key=lambda kv: kv[0])]
    gathered = [None] * world if rank == dst else None
    dist.gather_object(mine, gathered, dst=dst, group=host_group())
    if rank != dst:
        return None
    merged = sorted((kv for part in gathered for kv in part), key=lambda kv: kv[0])
    idx = [k for k, _ in merged]
    assert idx == list(range(len(idx))), "shards do not cover the input exactly once"
    return [p for _, p in merged]


def basecall_sharded(basecall_fn, model, reads, **kwargs):
    """Run `basecall_fn(model, reads_of_this_rank, **kwargs)` on every rank and return a generator over all
    (read_id, result) pairs in input order on rank 0 (an empty iterator elsewhere). `reads` is the FULL read iterable
    (sharded here, lazily); callers that can shard at the source (`Reader.get_reads(rank=, world=)`) should do that and use
    `ordered_records` directly."""
    rank, world, _ = env_rank_world()
    mine = (read for _, read in shard(reads, rank, world))
    out = ((getattr(read, "read_id", None), res) for read, res in basecall_fn(model, mine, **kwargs))
    return ordered_records(out, rank, world)
