#!/usr/bin/env python3
"""Workload for rocprofv3: a few steps of the bench hot path (hac-shaped model, batch 512 x chunk 10000).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python tools/profile_step.py [--decoder beam] [--model hac]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bonito_amd import decode, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="hac")
ap.add_argument("--decoder", default="beam")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--chunk", type=int, default=10000)
ap.add_argument("--quantize", action="store_true")
ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE", help="library option (bh_set_option), e.g. lstm_q8_variant=2")
a = ap.parse_args()
for kv in a.set:
    k, _, v = kv.partition("=")
    decode.set_option(k, int(v))
if a.model == "sup":
    a.batch, a.chunk = (256, 12000) if (a.batch, a.chunk) == (512, 10000) else (a.batch, a.chunk)
    model = synthetic.make_transformer_model(head_gain=4.0, batchsize=a.batch, chunksize=a.chunk)
else:
    model = synthetic.make_model(a.model, batchsize=a.batch, chunksize=a.chunk)
model.use_koi(batchsize=a.batch, chunksize=a.chunk, quantize=a.quantize)
model = model.half().cuda()
sig = torch.randn(a.batch, 1, a.chunk, device="cuda").half()
for _ in range(a.steps):
    sc = model(sig)
    if a.decoder == "beam":
        decode.beam_search(sc)
    else:
        decode.viterbi(sc)
torch.cuda.synchronize()
model._hip.check()
print("profiled", a.steps, "steps")
