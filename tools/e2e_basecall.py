#!/usr/bin/env python3
"""End-to-end throughput of the product path on synthetic reads, host work included: chunking, batching, H2D, encoder,
decode, D2H, stitching, formatting (FASTQ with move tables) and -- with several ranks -- the record merge on rank 0.

    python tools/e2e_basecall.py [--model hac|fast] [--reads 30000] [--mean-len 100000] [--devices 0-7] [--reps 2]

`--devices`: one process per listed GPU (a device may be listed twice), reads sharded round-robin, every rank formats
its own records, rank 0 merges them in input order and writes them (to /dev/null) -- the same code path as
`python -m bonito_amd basecaller --devices ...`, so host-feed and writer limits of an N-GPU node show up here
(bench.py keeps its batches resident on the device and cannot see them).
"""
import argparse
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # as the CLI does (several lanes need more than four hardware queues)
import subprocess
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="hac", choices=["hac", "fast"])
    ap.add_argument("--reads", type=int, default=30000)
    ap.add_argument("--mean-len", type=int, default=100000)
    ap.add_argument("--devices", default=None)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--quantize", action="store_true", help="the 8-bit recurrent path (two lanes, calls of 1024 chunks: the automatic choice)")
    ap.add_argument("--batchsize", type=int, default=512, help="chunks per engine call (more than 512: the recurrent kernels pair rings)")
    return ap.parse_args()


def launch(a):
    from bonito_amd.cli.basecaller import parse_devices
    devices = parse_devices(a.devices)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    argv, skip = [], False
    for tok in sys.argv[1:]:
        if skip:
            skip = False
        elif tok == "--devices":
            skip = True
        elif not tok.startswith("--devices="):
            argv.append(tok)
    procs = []
    for rank, dev in enumerate(devices):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(len(devices)), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(len(devices)),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIP_VISIBLE_DEVICES=str(dev))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    return 1 if any(p.wait() for p in procs) else 0


def main():
    a = parse()
    if a.devices and "RANK" not in os.environ:
        sys.exit(launch(a))
    import numpy as np
    import torch.distributed as dist
    from bonito_amd import io as bio
    from bonito_amd import parallel, synthetic, util
    import importlib
    basecall_records = importlib.import_module("bonito_amd.crf.basecall").basecall_records

    rank, world, _ = parallel.env_rank_world()
    if world > 1:
        parallel.init("gloo")
    util.limit_host_threads(8)
    model = synthetic.make_model(a.model, batchsize=a.batchsize, chunksize=10000)
    model.use_koi(batchsize=a.batchsize, chunksize=9996, quantize=a.quantize)
    model = model.half().cuda()

    class Read:
        run_id, filename, channel, mux, start, duration, template_start, template_duration, trimmed_samples = "run", "f", 0, 0, 0.0, 0.0, 0.0, 0.0, 0

        def __init__(self, i, sig):
            self.read_id, self.signal, self.num_samples = "read_%d" % i, sig, len(sig)

    rng = np.random.default_rng(1)
    lens = np.clip(rng.normal(a.mean_len, a.mean_len / 3, a.reads), 5000, None).astype(int)
    # a pool of distinct signals reused cyclically (30 000 reads of 1e5 samples would be 12 GB of float32): every read is a window
    # of one of them, with its own id and length
    pool = [np.random.default_rng(100 + k).standard_normal(int(lens.max()) + 1).astype(np.float32) for k in range(16)]

    def my_reads():
        for i, n in enumerate(lens):
            if i % world == rank:
                yield Read(i, pool[i % len(pool)][:int(n)])

    total = int(lens.sum())
    for rep in range(a.reps):
        if world > 1:
            dist.barrier(group=parallel.host_group())
        t0 = time.perf_counter()
        records = basecall_records(model, my_reads(), "fastq", chunksize=9996, overlap=498, batchsize=a.batchsize)
        records = parallel.ordered_records(records, rank, world)
        if rank == 0:
            with open(os.devnull, "w") as sink:
                w = bio.Writer("fastq", records, fd=sink, preformatted=True)
                w.start()
                w.join()
            if w.error is not None:
                raise w.error
            done = sum(n for _, n in w.log)
        if world > 1:
            dist.barrier(group=parallel.host_group())
        dt = time.perf_counter() - t0
        if rank == 0:
            assert done == total, (done, total)
            print("%s x%d rank(s), host cpus %s: %d reads, %.3e samples, %.2f s -> %.3e samples/s end to end (%.3e per rank)"
                  % (a.model, world, sorted(os.sched_getaffinity(0))[:4] if len(os.sched_getaffinity(0)) <= 4 else len(os.sched_getaffinity(0)),
                     a.reads, total, dt, total / dt, total / dt / world), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
