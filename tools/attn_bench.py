#!/usr/bin/env python3
"""The windowed attention kernel alone (bh_attention_prerotated): correctness against a torch restatement on the device, then its time
at the bench's call shapes.  usage (GPU box): python tools/attn_bench.py [--set attn_waves=8]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bonito_amd import _lib, decode

ap = argparse.ArgumentParser()
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
for kv in a.set:
    k, _, v = kv.partition("=")
    decode.set_option(k, int(v))
lib = _lib.lib()
dev = torch.device("cuda", 0)
H, HD = 8, 64
D = H * HD


def run(qkv, N, T, wl, wr):
    out = torch.empty((N * T, D), dtype=torch.float16, device=dev)
    _lib.check(lib.bh_attention_prerotated(_lib.ptr(qkv), _lib.ptr(out), N, T, H, HD, wl, wr, _lib.stream_ptr(dev)), "bh_attention_prerotated")
    return out


def reference(qkv, N, T, wl, wr):
    x = qkv.float().view(N, T, 3, H, HD)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * 0.6931471805599453             # q carries log2(e) / sqrt(d): scores are in base-2 units
    i = torch.arange(T, device=dev)[:, None]
    j = torch.arange(T, device=dev)[None, :]
    s = s.masked_fill(~((j >= i - wl) & (j <= i + wr)), float("-inf"))
    return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(N * T, D)


gen = torch.Generator(device=dev).manual_seed(3)
for N, T, wl, wr in ((2, 1000, 127, 128), (3, 333, 127, 128), (2, 1667, 127, 128), (2, 200, 40, 17), (1, 17, 127, 128), (2, 1000, 128, 128)):
    qkv = (torch.randn(N * T, 3 * D, generator=gen, device=dev) * 0.7).half()
    got = run(qkv, N, T, wl, wr).float()
    want = reference(qkv, N, T, wl, wr)
    err = (got - want).abs().max().item()
    print("check N=%d T=%4d window (%d, %d): max|d| %.2e %s" % (N, T, wl, wr, err, "ok" if err < 4e-3 else "WRONG"), flush=True)
for N, T in ((512, 1000), (512, 1667), (256, 1000)):
    qkv = (torch.randn(N * T, 3 * D, generator=gen, device=dev) * 0.7).half()
    for _ in range(3):
        run(qkv, N, T, 127, 128)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
    ev[0].record()
    for r in range(a.reps):
        run(qkv, N, T, 127, 128)
        ev[r + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(a.reps))
    flop = 2.0 * 2 * T * 256 * D * N
    print("time  N=%d T=%4d: median %.4f ms, min %.4f (%.0f TFLOP/s at the median; per 256 chunks %.4f ms)"
          % (N, T, ms[len(ms) // 2], ms[0], flop / ms[len(ms) // 2] / 1e9, ms[len(ms) // 2] * 256 / N), flush=True)
