#!/usr/bin/env python3
"""A/B of LSTM exchange knobs on the hac shape (timing only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bonito_amd import synthetic
model = synthetic.make_model("hac")
model.use_koi(batchsize=512, chunksize=10000, quantize=False)
model = model.half().cuda()
sig = torch.randn(512, 1, 10000, device="cuda").half()
ref = model(sig)
opt = os.environ.get("OPT", "lstm_tune")           # any encoder option taking 0 / 1, e.g. OPT=lstm_prefill
for tune in [int(v) for v in os.environ.get("VALS", "0,1,0,1").split(",")]:
    model._hip.set_option(opt, tune)
    for _ in range(2):
        out = model(sig)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        out = model(sig)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    model._hip.check()
    print(opt + "=%d  encoder %.2f ms  identical=%s" % (tune, dt * 1e3, bool(torch.equal(out, ref))))
