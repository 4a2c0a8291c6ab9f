#!/usr/bin/env python3
"""End-to-end throughput of bonito_amd.crf.basecall (host chunking/batching/stitching included) on synthetic reads.
    python tools/e2e_basecall.py [hac|fast] [n_reads] [mean_len]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bonito_amd import synthetic, util
from bonito_amd.crf import basecall

name = sys.argv[1] if len(sys.argv) > 1 else "hac"
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
mean_len = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
util.limit_host_threads(8)
model = synthetic.make_model(name, batchsize=512, chunksize=10000)
model.use_koi(batchsize=512, chunksize=9996, quantize=False)
model = model.half().cuda()


class Read:
    def __init__(self, i, sig):
        self.read_id, self.signal = "read_%d" % i, sig


rng = np.random.default_rng(1)
lens = np.clip(rng.normal(mean_len, mean_len / 3, n_reads), 5000, None).astype(int)
reads = [Read(i, rng.standard_normal(int(n)).astype(np.float32)) for i, n in enumerate(lens)]
total = int(lens.sum())
for rep in range(2):
    t0 = time.perf_counter()
    nb = 0
    for read, res in basecall(model, reads, chunksize=9996, overlap=498, batchsize=512):
        nb += len(res["sequence"])
    dt = time.perf_counter() - t0
    print("%s: %d reads, %.3e samples, %.2f s -> %.3e samples/s end to end (%d bases)" % (name, n_reads, total, dt, total / dt, nb))
