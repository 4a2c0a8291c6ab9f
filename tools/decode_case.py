#!/usr/bin/env python3
"""Decode-stage timing on real encoder output vs random scores (run under rocprofv3 --kernel-trace for per-kernel times).
    python tools/decode_case.py fast|hac|random256|random1024 [viterbi]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bonito_amd import decode, synthetic
case = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else "beam"
N, L = 512, 10000
if case.startswith("random"):
    C = int(case[6:])
    scores = (torch.randn(N, 1667, C, device="cuda") * 2.5).clamp(-5, 5).half()
else:
    model = synthetic.make_model(case)
    model.use_koi(batchsize=N, chunksize=L, quantize=False)
    model = model.half().cuda()
    scores = model(torch.randn(N, 1, L, device="cuda").half())
    torch.cuda.synchronize()
    print("scores: mean %.3f std %.3f max %.3f frac>=4 %.4f" % (scores.float().mean(), scores.float().std(), scores.float().max(),
                                                               (scores.float() >= 4).float().mean()))
dec = decode.CRFDecoder(N, scores.shape[1], scores.shape[2], "cuda:0", mode=mode)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    tk = dec.submit(scores)
    ev[1].record(); 
    r = tk.result()
    torch.cuda.synchronize()
    print("%s %s rep %d: host %.2f ms, device %.2f ms" % (case, mode, rep, (time.perf_counter() - t0) * 1e3, ev[0].elapsed_time(ev[1])))
seq = r["sequence"] if isinstance(r, dict) else r[0]
print("emitted bases per chunk: %.1f" % ((seq != 0).sum().item() / N))
