#!/usr/bin/env python3
"""Full-scale exactness / determinism check of the 8-bit recurrent kernel: N = 512 chunks (32 rings), both geometries.
Every step's int32 sums must equal integer matmuls of the quantised weights with the quantised input / with the int8 h the
kernel itself published one step earlier. Reports where the first mismatch is (step, ring, slice)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from bonito_amd import _lib
from oracle import lstm_q8_ref
from test_gpu_q8 import run_layer

H, N, T = 384, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 48
rng = np.random.default_rng(1)
w_ih = (rng.standard_normal((4 * H, H)) * 0.08).astype(np.float32)
w_hh = (rng.standard_normal((4 * H, H)) * 0.08).astype(np.float32)
bias = (rng.standard_normal(4 * H) * 0.3).astype(np.float32)
x = torch.from_numpy(np.clip(rng.standard_normal((T, N, H)) * 0.6, -1.2, 1.2).astype(np.float16))
q_ih, _ = lstm_q8_ref.quantise_rows(w_ih)
q_hh, _ = lstm_q8_ref.quantise_rows(w_hh)
xq = lstm_q8_ref.quantise_act(x.float().numpy(), 1.0).astype(np.float32)
want_x = (xq.reshape(T * N, H) @ q_ih.astype(np.float32).T).astype(np.int32).reshape(T, N, 4 * H)      # |sum| < 2^24: exact in fp32
ref = {}
for variant in (0, 1):
    for reverse in (0, 1):
        for rep in range(2):
            t0 = time.time()
            h16, hq, sums = run_layer(x, w_ih, w_hh, bias, 1.0, reverse, variant)
            dt = time.time() - t0
            okx = np.array_equal(sums[..., 0], want_x)
            order = list(range(T - 1, -1, -1)) if reverse else list(range(T))
            prev = np.zeros((N, H), np.float32)
            bad = None
            for k, t in enumerate(order):
                want_h = (prev @ q_hh.astype(np.float32).T).astype(np.int32)
                if not np.array_equal(sums[t, :, :, 1], want_h):
                    idx = np.argwhere(sums[t, :, :, 1] != want_h)
                    n, r = idx[0]
                    bad = (k, t, int(n) // 16, int(r) % H, len(idx))
                    break
                prev = hq[t].astype(np.float32)
            key = (variant, reverse)
            same = None
            if key in ref:
                same = bool(np.array_equal(ref[key], hq))
            ref.setdefault(key, hq)
            print("variant %d reverse %d rep %d: %.2fs x-sums exact %s, first bad h-sum (step, t, ring, unit, count) %s, same as rep 0: %s"
                  % (variant, reverse, rep, dt, okx, bad, same), flush=True)
print("variants agree (fwd):", np.array_equal(ref[(0, 0)], ref[(1, 0)]), "(rev):", np.array_equal(ref[(0, 1)], ref[(1, 1)]))
