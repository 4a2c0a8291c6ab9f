"""N > 1 path on CPU: world_size-2 (and 3) gloo processes shard reads at the record level, basecall them independently
(no collective on the data path), format their own records, and rank 0's REAL Writer streams the merged records in input
order -- byte-identical to the single-process output. Plus the timing MAX-reduce used by bench.py."""
import io
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bonito_amd import io as bio
from bonito_amd import parallel, reader


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


N_READS = 23


def make_reads_dir(path):
    rng = np.random.default_rng(7)
    for i in range(N_READS):
        n = int(rng.integers(900, 6000))
        np.save(os.path.join(path, "read%02d.npy" % i), (rng.standard_normal(n) * 12 + 90).astype(np.float32))
        with open(os.path.join(path, "read%02d.json" % i), "w") as fh:
            json.dump({"read_id": "rid-%02d" % i, "run_id": "run%d" % (i % 2), "channel": i}, fh)


def fake_basecall(model, reads, stride=6):
    """Stands in for bonito_amd.crf.basecall on CPU: same (read, dict) protocol, deterministic, content depends on the
    read's signal; some reads are empty / low quality so that the writer's filters are exercised."""
    for read in reads:
        T = len(read.signal) // stride
        h = int(np.abs(read.signal[:50]).sum() * 1000) % 9973
        rng = np.random.default_rng(h)
        moves = (rng.random(T) < 0.4).astype(np.int8)
        if h % 7 == 0:
            moves[:] = 0                                   # empty sequence -> skipped by the writer, still logged
        n = int(moves.sum())
        seq = "".join("ACGT"[b] for b in rng.integers(0, 4, n))
        q = 33 + (3 if h % 5 == 0 else 25)                 # q 3 reads fall under --min-qscore
        yield read, {"sequence": seq, "qstring": chr(q) * n, "moves": moves, "stride": stride}


def _run_writer(records_or_results, mode, preformatted, summary_path):
    buf = io.StringIO()
    w = bio.Writer(mode, records_or_results, fd=buf, min_qscore=7.0, summary_path=summary_path, preformatted=preformatted)
    w.start()
    w.join()
    if w.error is not None:
        raise w.error
    with open(summary_path) as fh:
        return buf.getvalue(), fh.read(), w.log


def _worker(rank, world, port, rdir, mode, batch, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = parallel.init("gloo")
    assert (r, w) == (rank, world)
    reads = reader.Reader(rdir).get_reads(rank=rank, world=world)           # record-level shard: only my files are loaded
    records = parallel.ordered_records(parallel.format_stream(fake_basecall(None, reads), mode, 7.0), rank, world, batch=batch)
    out = None
    if rank == 0:
        out = _run_writer(records, mode, True, os.path.join(rdir, "summary_w%d.tsv" % world))
    else:
        assert list(records) == []
    slow = parallel.max_over_ranks(1.0 + rank)
    dist.barrier()
    q.put((rank, out, slow))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,batch", [(2, "fastq", 4), (3, "sam", 64), (2, "fasta", 1)])
def test_multi_rank_streaming_writer_is_byte_identical(tmp_path, world, mode, batch):
    rdir = str(tmp_path)
    make_reads_dir(rdir)
    want = _run_writer(fake_basecall(None, reader.Reader(rdir).get_reads()), mode, False, os.path.join(rdir, "summary_1.tsv"))
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, rdir, mode, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict()
    for _ in range(world):
        rank, out, slow = q.get(timeout=180)
        outs[rank] = (out, slow)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    text, summary, log = outs[0][0]
    if mode == "sam":       # the @PG line quotes the command line of the writing process
        strip = lambda s: "\n".join(l for l in s.split("\n") if not l.startswith("@PG"))
        assert strip(text) == strip(want[0])
    else:
        assert text == want[0]                               # same bytes as the single-process run, in input order
    assert summary == want[1]
    assert log == want[2] and len(log) == N_READS            # every read logged once (filtered ones too), post-trim samples
    assert len(text) > 1000
    assert all(outs[r][1] == float(world) for r in range(world))     # MAX over ranks


def _skew_worker(rank, world, port, n, slow_rank, q):
    """Rank `slow_rank` produces every record ~40x slower than the others and with 10x the payload; one rank pauses for a while in
    the middle. Tiny windows (batch 2, window 2), so every back-pressure path is taken many times."""
    import time
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init("gloo")
    produced = []

    def records():
        for k, i in enumerate(range(rank, n, world)):
            if rank == slow_rank:
                time.sleep(0.004)
            if rank == (slow_rank + 1) % world and k == 20:
                time.sleep(0.5)                                    # a rank that sleeps
            produced.append(i)
            yield ("rec-%d" % i) * (10 if rank == slow_rank else 1), [i], ("id%d" % i, i)

    out = parallel.ordered_records(records(), rank, world, batch=2, window=2)
    got = list(out)
    dist.barrier()
    q.put((rank, got if rank == 0 else len(produced)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,slow", [(3, 1), (2, 0), (4, 3)])
def test_merge_with_a_slow_and_a_sleeping_rank_neither_deadlocks_nor_reorders(world, slow):
    """Skewed ranks (one with 10x the bytes per record and 40x the time, one that pauses): the merged stream is still every record
    once, in global order, and the run ends - the fast ranks wait in their bounded windows, nobody holds records to the end."""
    n = 151
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_skew_worker, args=(r, world, port, n, slow, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in outs[0]] == [[i] for i in range(n)]
    assert all(r[0] == ("rec-%d" % r[1][0]) * (10 if r[1][0] % world == slow else 1) for r in outs[0])
    assert sum(outs[r] for r in range(1, world)) == n - len(range(0, n, world))


def test_prefetch_propagates_producer_errors_and_bounds_the_queue():
    def boom():
        yield 1
        yield 2
        raise ValueError("producer failed")

    it = iter(parallel._Prefetch(boom(), 4))
    assert next(it) == 1 and next(it) == 2
    with pytest.raises(ValueError):
        next(it)
    produced = []

    def counted():
        for i in range(100):
            produced.append(i)
            yield i

    pf = parallel._Prefetch(counted(), 3)
    import time
    time.sleep(0.3)
    assert len(produced) <= 5                       # 3 queued + one in hand (+ one being produced): the producer waits
    assert list(pf) == list(range(100))


def test_reader_shard_is_a_partition_and_loads_only_its_own(tmp_path, monkeypatch):
    rdir = str(tmp_path)
    make_reads_dir(rdir)
    full = [r.read_id for r in reader.Reader(rdir).get_reads()]
    loaded = []
    real_load = np.load
    monkeypatch.setattr(np, "load", lambda p, *a, **k: (loaded.append(os.path.basename(p)), real_load(p, *a, **k))[1])
    parts = [[r.read_id for r in reader.Reader(rdir).get_reads(rank=k, world=4)] for k in range(4)]
    assert sorted(sum(parts, [])) == sorted(full) and len(loaded) == len(full)      # each file opened by exactly one rank
    for k in range(4):
        assert parts[k] == full[k::4]
    # n_max and read_ids are applied before sharding: the union over ranks equals the unsharded selection
    sel = [r.read_id for r in reader.Reader(rdir).get_reads(n_max=10, read_ids={"rid-03"}, skip=True)]
    got = [[r.read_id for r in reader.Reader(rdir).get_reads(n_max=10, read_ids={"rid-03"}, skip=True, rank=k, world=3)]
           for k in range(3)]
    assert len(sel) == 10 and "rid-03" not in sel
    for k in range(3):
        assert got[k] == sel[k::3]


def test_shard_is_a_partition():
    items = list(range(23))
    parts = [list(parallel.shard(items, r, 4)) for r in range(4)]
    flat = sorted(kv for p in parts for kv in p)
    assert flat == [(i, i) for i in items]
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_paths_are_identity():
    assert parallel.gather_in_order([(2, "c"), (0, "a"), (1, "b")], 0, 1) == ["a", "b", "c"]
    assert list(parallel.ordered_records(iter("abc"), 0, 1)) == ["a", "b", "c"]
    assert parallel.max_over_ranks(3.5) == 3.5


def test_parse_devices():
    from bonito_amd.cli.basecaller import parse_devices
    assert parse_devices("0-7") == list(range(8)) and parse_devices("0,2,5") == [0, 2, 5] and parse_devices("1-2,0") == [1, 2, 0]
    with pytest.raises(ValueError):
        parse_devices("")


def test_child_device_mask_follows_the_parent_masks():
    """`--devices` indices are relative to what the launcher may see. HIP numbers devices inside the set ROCR_VISIBLE_DEVICES
    leaves, and the ROCR mask stays in the child's environment (advisor finding, round 3)."""
    from bonito_amd.cli.basecaller import child_device_mask
    assert child_device_mask(3, {}) == "3"
    assert child_device_mask(1, {"HIP_VISIBLE_DEVICES": "4,6,7"}) == "6"
    # ROCR-only mask: HIP indices 0..1 address the filtered set; the physical ids 2 / 3 would see nothing
    assert [child_device_mask(d, {"ROCR_VISIBLE_DEVICES": "2,3"}) for d in (0, 1)] == ["0", "1"]
    # both: the HIP mask's values already are indices into the ROCR set
    assert child_device_mask(1, {"ROCR_VISIBLE_DEVICES": "2,3,5", "HIP_VISIBLE_DEVICES": "2,0"}) == "0"
    for env in ({"HIP_VISIBLE_DEVICES": "4,6"}, {"ROCR_VISIBLE_DEVICES": "2,3"}):
        with pytest.raises(SystemExit):
            child_device_mask(2, env)


def _dying_worker(rank, world, port, rdir, mode, victim, after, q):
    """Rank `victim` dies (os._exit, no goodbye) after producing `after` records; rank 0 re-does the rest of its shard (rescue)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init("gloo")
    rd = reader.Reader(rdir)

    def records_of(r, skip=0):
        import itertools
        reads = itertools.islice(rd.get_reads(rank=r, world=world), skip, None)
        for k, rec in enumerate(parallel.format_stream(fake_basecall(None, reads), mode, 7.0)):
            if rank == victim and r == rank and skip + k == after:
                os._exit(3)
            yield rec

    lost = []
    records = parallel.ordered_records(records_of(rank), rank, world, batch=2, window=2, rescue=lambda r, k: records_of(r, k),
                                       on_rank_lost=lambda r, k, exc: lost.append((r, k)))
    out = None
    if rank == 0:
        out = _run_writer(records, mode, True, os.path.join(rdir, "summary_dying.tsv"))
    else:
        assert list(records) == []
    q.put((rank, out, lost))


@pytest.mark.parametrize("world,victim,after", [(3, 1, 3), (3, 2, 0), (2, 1, 5)])
def test_a_rank_that_dies_is_replaced_by_rank_0_and_the_output_is_unchanged(tmp_path, world, victim, after):
    """SURVEY 5 "failure detection" (re-queue on another replica): a worker is killed in the middle of its stream; rank 0 keeps the
    records it had received from it, produces the rest of that rank's shard itself and writes the same bytes as a one-rank run; the
    surviving ranks finish normally (closing message instead of a barrier that would wait for the dead)."""
    rdir = str(tmp_path)
    make_reads_dir(rdir)
    mode = "fastq"
    want = _run_writer(fake_basecall(None, reader.Reader(rdir).get_reads()), mode, False, os.path.join(rdir, "summary_1.tsv"))
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dying_worker, args=(r, world, port, rdir, mode, victim, after, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {}
    for _ in range(world - 1):                                # the victim never reports
        rank, out, lost = q.get(timeout=180)
        outs[rank] = (out, lost)
    for r, p in enumerate(procs):
        p.join(timeout=60)
        assert p.exitcode == (3 if r == victim else 0), (r, p.exitcode)
    text, summary, log = outs[0][0]
    assert text == want[0] and summary == want[1] and log == want[2]
    (lost_rank, k), = outs[0][1]
    assert lost_rank == victim and k <= after                 # what had arrived is kept; the rest was re-done from record k on


def test_packed_record_blocks_round_trip_and_match_the_writer_bytes(tmp_path):
    """Round 6: records travel from the ranks to rank 0 as packed blocks (one buffer per message, summary rows rendered on the producing
    rank with the csv dialect of io.Writer) instead of pickles. Unpacking gives back the same texts / rows / log entries, and a Writer fed
    the unpacked (bytes) records writes the very bytes it writes for the original (str, list) records - into a real file (binary layer)
    and into a StringIO (decode path)."""
    rdir = str(tmp_path)
    make_reads_dir(rdir)
    recs = list(parallel.format_stream(fake_basecall(None, reader.Reader(rdir).get_reads()), "sam", 7.0))
    assert any(t is None for t, _, _ in recs) and sum(t is not None for t, _, _ in recs) > 5          # filtered reads travel too (log only)
    blob = parallel.pack_records(recs, True)
    got, last = parallel.unpack_records(memoryview(bytes(blob)))
    assert last and len(got) == len(recs)
    for (t, r, l), (t2, r2, l2) in zip(recs, got):
        assert l2 == (l[0], l[1])
        assert (t is None) == (t2 is None) and (t is None or bytes(t2).decode() == t)
        assert (r is None) == (r2 is None) and (r is None or r2 == parallel.render_summary_row(r))
    assert parallel.unpack_records(parallel.pack_records([], False)) == ([], False)
    with pytest.raises(ValueError):
        parallel.unpack_records(bytes(64))
    want = _run_writer(iter(recs), "sam", True, os.path.join(rdir, "s_a.tsv"))
    have = _run_writer(iter(got), "sam", True, os.path.join(rdir, "s_b.tsv"))                        # StringIO sink: the decode path
    strip = lambda s: "\n".join(ln for ln in s.split("\n") if not ln.startswith("@PG"))
    assert strip(have[0]) == strip(want[0]) and have[1] == want[1] and have[2] == want[2]
    path = os.path.join(rdir, "out.sam")
    mixed = [recs[i] if i % 2 else got[i] for i in range(len(recs))]                                  # rank 0's own str records between the peers' bytes
    with open(path, "w") as fh:
        w = bio.Writer("sam", iter(mixed), fd=fh, summary_path=os.path.join(rdir, "s_c.tsv"), preformatted=True)
        w.start(); w.join()
        assert w.error is None
    assert strip(open(path).read()) == strip(want[0]) and open(os.path.join(rdir, "s_c.tsv")).read() == want[1]


def _packed_worker(rank, world, port, rdir, mode, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init("gloo")
    reads = reader.Reader(rdir).get_reads(rank=rank, world=world)
    records = parallel.ordered_records(parallel.format_stream(fake_basecall(None, reads), mode, 7.0), rank, world, batch=3, window=2, packed=True)
    out = None
    if rank == 0:
        out = _run_writer(records, mode, True, os.path.join(rdir, "summary_p%d.tsv" % world))
    else:
        assert list(records) == []
    q.put((rank, out))
    parallel.shutdown()


@pytest.mark.parametrize("world,mode", [(2, "fastq"), (4, "sam")])
def test_multi_rank_packed_streams_are_byte_identical(tmp_path, world, mode):
    rdir = str(tmp_path)
    make_reads_dir(rdir)
    want = _run_writer(fake_basecall(None, reader.Reader(rdir).get_reads()), mode, False, os.path.join(rdir, "summary_1.tsv"))
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_packed_worker, args=(r, world, port, rdir, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    text, summary, log = outs[0]
    strip = lambda s: "\n".join(ln for ln in s.split("\n") if not ln.startswith("@PG"))
    assert strip(text) == strip(want[0]) and summary == want[1] and log == want[2] and len(log) == N_READS
