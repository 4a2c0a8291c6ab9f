#!/usr/bin/env python3
"""Can the HOST side of an 8-GPU node keep eight ranks fed? (the scaling risk of a shard-by-read basecaller is the Python
host, not xGMI.) No GPU needed: R worker processes (one per would-be rank, like `basecaller --devices 0-7`) each run the PRODUCT host
pipeline - `basecall_records`: chunking + fp16 batch assembly (`chunk_batches`, `bh_host_chunk_rows`), the three pipeline threads,
one `bh_host_format_read` call per read, FASTQ text with move tables into a sink - with the two device stages replaced IN THIS TOOL by
stubs: `encode` returns at once, `decode` sleeps for the time a real engine call takes (56 ms per 2048-chunk call of the hac model at
13.7 ms per 512-chunk batch, scaled by the chunks in the call) and hands back synthetic int8 planes of the density of real calls.
What is measured is what one rank's host threads sustain while seven others compete for the same cores:

    taskset -c 0-15 python tools/host_scale.py --ranks 8          # the 16-core quota of a GPU box
    python tools/host_scale.py --ranks 1                          # one rank alone, for comparison

A rank keeps up with its GPU if its rate here is at the device rate of the stub (2048 * 9996 samples / 56 ms = 3.66e8 samples/s):
then the host is not the bottleneck. Nothing of the product path is changed or stubbed outside this process.
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(a):
    import importlib
    import numpy as np
    import torch
    from bonito_amd import synthetic, util
    bc = importlib.import_module("bonito_amd.crf.basecall")
    util.limit_host_threads(2)
    chunksize, overlap = 9996, 498
    model = synthetic.make_model("hac", batchsize=a.batchsize, chunksize=10000)
    stride = model.stride
    T = chunksize // stride
    rng = np.random.default_rng(7 + (0 if a.merge else a.rank))          # --merge: every rank sees the SAME read set and takes its shard
    per_call = bc.batches_per_call(model, a.batchsize, False, chunksize, 1)
    call = a.batchsize * per_call
    mv = (rng.random((call, T)) < 0.42).astype(np.int8)
    seq = np.where(mv != 0, np.array([65, 67, 71, 84], np.int8)[rng.integers(0, 4, (call, T))], 0).astype(np.int8)
    qs = np.where(mv != 0, rng.integers(36, 75, (call, T)).astype(np.int8), 0).astype(np.int8)
    planes = torch.from_numpy(np.stack([seq, qs, mv]))

    class StubPipeline:
        """the two device stages of crf/basecall.py::_Pipeline, as sleeps (tool only)"""
        lanes = 1

        def __init__(self, *args, **kw):
            pass

        def encode(self, batch):
            return (int(batch.shape[0]),)

        def decode(self, n):
            time.sleep(a.call_ms * 1e-3 * n / 2048.0)
            return planes[:, :n]

    bc._Pipeline = StubPipeline

    class Read:
        run_id, filename, channel, mux, start, duration, template_start, template_duration, trimmed_samples = "run", "f", 0, 0, 0.0, 0.0, 0.0, 0.0, 0

        def __init__(self, i, sig):
            self.read_id, self.signal, self.num_samples = "read_%d" % i, sig, len(sig)

    lens = np.clip(rng.normal(a.mean_len, a.mean_len / 3, a.reads), 5000, None).astype(int)
    pool = [np.random.default_rng(100 + k).standard_normal(int(lens.max()) + 1).astype(np.float32) for k in range(4)]
    if a.merge:
        # the PRODUCT's multi-process path (round 6): every rank produces the records of ITS shard (read i belongs to rank i % world, as
        # Reader.get_reads(rank=, world=) shards), rank 0 merges the streams in input order through parallel.ordered_records and is the
        # only writer (bonito_amd.io.Writer into /dev/null) - what `basecaller --devices 0-7` does around the device stages
        from bonito_amd import io as bio
        from bonito_amd import parallel
        os.environ.update(RANK=str(a.rank), WORLD_SIZE=str(a.ranks), LOCAL_RANK=str(a.rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port))
        parallel.init("gloo")
        mine = [(i, n) for i, n in enumerate(lens) if i % a.ranks == a.rank]
        reads = (Read(i, pool[i % 4][:int(n)]) for i, n in mine)
        records = bc.basecall_records(model, reads, "fastq", chunksize=chunksize, overlap=overlap, batchsize=a.batchsize)
        t0 = time.perf_counter()
        merged = parallel.ordered_records(records, a.rank, a.ranks, packed=not a.pickled)
        if a.rank == 0:
            with open(os.devnull, "w") as sink:
                w = bio.Writer("fastq", merged, fd=sink, preformatted=True)
                w.start()
                w.join()
            if w.error is not None:
                raise w.error
            dt = time.perf_counter() - t0
            total = sum(n for _, n in w.log)
            assert len(w.log) == a.reads and total == int(lens.sum()), (len(w.log), total)
            print("merge: %d ranks, %d reads, %.3e samples, %.2f s -> %.3e samples/s through rank 0's ordered writer (%.3e per rank; stub device "
                  "rate %.3e per rank)" % (a.ranks, len(w.log), total, dt, total / dt, total / dt / a.ranks, 2048 * 9996 / (a.call_ms * 1e-3)), flush=True)
        else:
            assert list(merged) == []
        parallel.shutdown()
        return
    reads = (Read(i, pool[i % 4][:int(n)]) for i, n in enumerate(lens))
    t0 = time.perf_counter()
    n_out = n_bytes = 0
    with open(os.devnull, "w") as sink:
        for text, row, log in bc.basecall_records(model, reads, "fastq", chunksize=chunksize, overlap=overlap, batchsize=a.batchsize):
            if text is not None:
                sink.write(text)
                n_out += 1
                n_bytes += len(text)
    dt = time.perf_counter() - t0
    total = int(lens.sum())
    print("rank %d: %d reads, %.3e samples, %.2f s -> %.3e samples/s (stub device rate %.3e; %d-chunk calls; %.0f MB of FASTQ)"
          % (a.rank, n_out, total, dt, total / dt, 2048 * 9996 / (a.call_ms * 1e-3), call, n_bytes / 1e6), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--mean-len", type=int, default=100000)
    ap.add_argument("--batchsize", type=int, default=512)
    ap.add_argument("--call-ms", type=float, default=56.0, help="device time of one 2048-chunk engine call + decode (hac: 4 x 13.7 ms)")
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--merge", action="store_true",
                    help="the ranks share ONE read set (--reads in total, sharded round-robin) and rank 0 merges their record streams in input "
                         "order and writes (parallel.ordered_records + io.Writer): the aggregate the 8-GPU product run is bounded by on the host side")
    ap.add_argument("--port", type=int, default=0)
    ap.add_argument("--pickled", action="store_true", help="--merge with the pickled record messages of rounds 2-5 instead of packed blocks (A/B)")
    a = ap.parse_args()
    if a.rank >= 0:
        return worker(a)
    t0 = time.perf_counter()
    extra = []
    if a.merge:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            extra = ["--port", str(sock.getsockname()[1])]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r)] + sys.argv[1:] + extra, stdout=subprocess.PIPE, text=True)
             for r in range(a.ranks)]
    rates = []
    for p in procs:
        out = p.communicate()[0]
        sys.stdout.write(out)
        for line in out.splitlines():
            if "samples/s" in line:
                rates.append(float(line.split("->")[1].split()[0]))
    if a.merge:
        return
    cpus = sorted(os.sched_getaffinity(0))
    print("%d rank(s) on %d host cpu(s): sum %.3e samples/s, slowest rank %.3e, wall %.1f s" % (
        a.ranks, len(cpus), sum(rates), min(rates) if rates else 0.0, time.perf_counter() - t0))


if __name__ == "__main__":
    main()
