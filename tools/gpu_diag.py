#!/usr/bin/env python3
"""Prints parity error statistics + quick timings on the GPU box (used to set test tolerances)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from conftest import NN_FIXTURES, build_model, load_nn_fixture, ref_scores_to_koi
from bonito_amd.engine import HipEncoder
from bonito_amd import decode, synthetic
from oracle import crf_ref

print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count, "CUs")
for name in NN_FIXTURES:
  try:
    cfg, sd, x, y = load_nn_fixture(name)
    model = build_model(cfg, sd)
    enc = HipEncoder(model, batchsize=x.shape[0], chunksize=x.shape[-1])
    got = enc(x.half().cuda()); enc.check()
    has_blank = any(getattr(m, "blank_score", None) is not None for m in model.modules())
    want = ref_scores_to_koi(y, has_blank)
    d = (got.cpu().float() - want).abs()
    print("%-20s max %.4f mean %.5f  (|want| max %.2f)" % (name, d.max(), d.mean(), want.abs().max()))
  except Exception as exc:
    print(name, "FAILED", exc)

for name, N in (("fast", 512), ("hac", 512)):
    model = synthetic.make_model(name)
    model.use_koi(batchsize=N, chunksize=10000, quantize=False)
    model = model.half().cuda()
    sig = torch.randn(N, 1, 10000, device="cuda").half()
    for _ in range(2):
        sc = model(sig)
    model._hip.check()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sc = model(sig)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    mv, pa = decode.viterbi(sc)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    mv, pa = decode.viterbi(sc)
    mv_v = mv
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    model._hip.check()
    model._hip.profile(True)
    model(sig)
    prof = model._hip.profile_read()
    print(name, "encoder ms/batch %.2f  viterbi ms %.2f  samples/s enc-only %.3e" % ((t1 - t0) / 3 * 1e3, (t3 - t2) * 1e3, N * 10000 / ((t1 - t0) / 3)))
    print("   ", {k: (round(v[0], 3), v[1]) for k, v in prof.items() if v[1]})
    print("    scores finite:", bool(torch.isfinite(sc).all()), "std %.3f" % sc.float().std().item(),
          "bases/chunk %.1f" % float((pa != 0).sum() / N))
    om, op, _ = crf_ref.viterbi(sc[:2].cpu().numpy(), model.seqdist.state_len)
    print("    viterbi exact on 2 chunks:", np.array_equal(pa[:2].numpy(), op))
    try:
        decode.beam_search(sc)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        seq, qs, mv = decode.beam_search(sc)
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        oseq, oqs, omv, _ = crf_ref.beam_search(sc[:2].cpu().numpy(), model.seqdist.state_len)
        print("    beam ms %.2f  exact on 2 chunks: %s  bases/chunk %.1f  agree-with-viterbi %.4f" % (
            (t5 - t4) * 1e3, np.array_equal(seq[:2].numpy(), oseq) and np.array_equal(mv[:2].numpy(), omv),
            float(mv.sum() / N), float((mv == mv_v).float().mean())))
    except Exception as exc:
        print("    beam FAILED", exc)
