#!/usr/bin/env python3
"""Host-side cost of the consumer half of the product path, no GPU: decoded planes [3, N, T] int8 per engine call (synthetic, with the
density of real calls) -> unbatchify -> stitch -> strings -> FASTQ record with move table. Prints samples/s of ONE host thread.
    python tools/host_bench.py [--fused]     (--fused: the one-call-per-read C++ path, bonito_amd.crf.basecall.records_from_planes)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bonito_amd import io as bio, util
from bonito_amd.crf import basecall as _  # noqa
import importlib
bc = importlib.import_module("bonito_amd.crf.basecall")

ap = argparse.ArgumentParser()
ap.add_argument("--fused", action="store_true")
ap.add_argument("--reads", type=int, default=3000)
ap.add_argument("--mean-len", type=int, default=100000)
ap.add_argument("--batch", type=int, default=1024)
a = ap.parse_args()
chunksize, overlap, stride = 9996, 498, 6
T = chunksize // stride


class Read:
    run_id, filename, channel, mux, start, duration, template_start, template_duration, trimmed_samples = "run", "f", 0, 0, 0.0, 0.0, 0.0, 0.0, 0
    def __init__(self, i, n):
        self.read_id, self.num_samples, self.signal_len = "read_%d" % i, n, n
    signal = None


rng = np.random.default_rng(1)
lens = np.clip(rng.normal(a.mean_len, a.mean_len / 3, a.reads), 12000, None).astype(int)
reads = [Read(i, int(n)) for i, n in enumerate(lens)]
# chunk table like chunk_batches: keys per batch
def n_chunks(Tn):
    step = chunksize - overlap
    stub = (Tn - overlap) % step
    return (Tn - stub - chunksize) // step + 1 + (1 if stub > 0 else 0)
# one shared random planes batch (moves ~ 0.4 density, bases where moves)
mv = (rng.random((a.batch, T)) < 0.42).astype(np.int8)
seq = np.where(mv != 0, rng.integers(0, 4, (a.batch, T)).astype(np.int8) * 0 + np.array([65, 67, 71, 84], np.int8)[rng.integers(0, 4, (a.batch, T))], 0).astype(np.int8)
qs = np.where(mv != 0, rng.integers(36, 75, (a.batch, T)).astype(np.int8), 0).astype(np.int8)
planes = torch.from_numpy(np.stack([seq, qs, mv]))

def batches():
    keys, pos = [], 0
    for r in reads:
        n, lo = n_chunks(r.num_samples), 0
        key = (r, 0, r.num_samples)
        while lo < n:
            take = min(n - lo, a.batch - pos)
            keys.append((key, (pos, pos + take)))
            pos += take; lo += take
            if pos == a.batch:
                yield tuple(keys), planes
                keys, pos = [], 0
    if pos:
        yield tuple(keys), planes[:, :pos]

t0 = time.perf_counter()
n_out, n_bytes = 0, 0
if a.fused:
    for text, row, log in bc.records_from_planes(batches(), chunksize, overlap, stride, "fastq"):
        n_out += 1; n_bytes += len(text)
else:
    results = ((read, bc.fmt_planes(stride, bc.stitch_planes(sc, end - start, chunksize, overlap, stride, False), False))
               for ((read, start, end), sc) in util.unbatchify(batches(), dim=1))
    for read, res in results:
        text, row, log = bio.format_record(read, res, "fastq")
        n_out += 1; n_bytes += len(text)
dt = time.perf_counter() - t0
print("%s: %d reads, %.3e samples, %.2f s -> %.3e samples/s per host thread (%.1f us per read, %.1f MB of text)"
      % ("fused C++" if a.fused else "python", n_out, lens.sum(), dt, lens.sum() / dt, 1e6 * dt / n_out, n_bytes / 1e6))
