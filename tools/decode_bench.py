#!/usr/bin/env python3
"""Decode stage alone (bh_beam_search - or, with the word `viterbi`, bh_crf_viterbi - on resident scores, no D2H): milliseconds per call for
a few option settings.
    python tools/decode_bench.py [N T C] [viterbi] [name=value ...]      e.g.  2048 1667 1024 beam_cpw=4 / 2048 1667 1024 viterbi viterbi_quad=0"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bonito_amd import decode, _lib
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
opts = [a for a in sys.argv[1:] if "=" in a]
N, T, C = (nums + [2048, 1667, 1024])[:3] if len(nums) >= 3 else (2048, 1667, 1024)
torch.manual_seed(5)
sc = (torch.randn(N, T, C, device="cuda") * 2.5).clamp(-5, 5).half()
REPS = int(os.environ.get("BH_DECODE_BENCH_REPS", "8"))
sl = decode.state_len_of(C)
lib = _lib.lib()
ws = torch.empty(lib.bh_beam_search_workspace(N, T, sl), dtype=torch.uint8, device="cuda")
out = torch.empty((3, N, T), dtype=torch.int8, device="cuda")
VITERBI = "viterbi" in sys.argv[1:]
vws = torch.empty(lib.bh_crf_viterbi_workspace(N, T, sl), dtype=torch.uint8, device="cuda") if VITERBI else None
def run():
    if VITERBI:
        _lib.check(lib.bh_crf_viterbi(_lib.ptr(sc), N, T, sl, 0, 2.0, T * C, C, _lib.ptr(vws), _lib.ptr(out[0]), _lib.ptr(out[1]), None,
                                      _lib.stream_ptr("cuda:0")), "bh_crf_viterbi")
        return
    _lib.check(lib.bh_beam_search(_lib.ptr(sc), N, T, sl, 32, 100.0, 2.0, 1.0, 0.0, _lib.ptr(ws), _lib.ptr(out[0]), _lib.ptr(out[1]),
                                  _lib.ptr(out[2]), None, _lib.stream_ptr("cuda:0")), "bh_beam_search")
def timed(tag):
    for _ in range(min(3, REPS)): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): run()
    e1.record(); torch.cuda.synchronize()
    ref = out.clone()
    print("%-28s %d x %d x %d: %.3f ms per call (%.3f per 512 chunks)  checksum %d" % (tag, N, T, C, e0.elapsed_time(e1) / REPS, e0.elapsed_time(e1) / REPS * 512 / N,
          int(ref.to(torch.int64).sum())), flush=True)
timed("default")
for kv in opts:
    k, v = kv.split("=")
    decode.set_option(k, int(v))
    timed(kv)
