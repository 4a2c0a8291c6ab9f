#!/usr/bin/env python3
import os, sys
os.environ["BH_BEAM_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bonito_amd import decode, _lib
N, T, C = 512, 1667, 1024
g = torch.Generator(device="cuda").manual_seed(1)
for name, gain in (("flat random (bench-like)", 2.5), ("peaked", 6.0)):
    sc = (torch.randn(N, T, C, generator=g, device="cuda") * gain).clamp(-5, 5).half()
    dec = decode.CRFDecoder(N, T, C, "cuda:0", mode="beam")
    dec.submit(sc).result()
    dec.submit(sc).result()
    ws = dec.ws.cpu().numpy()
    al = lambda x: (x + 255) // 256 * 256
    S = 256
    off = al(N * (T + 1) * S * 4) + al(N * (T + 1) * 8) + al(N * 8) + al(N * T * 4 * 4) + al(N * T * 32) + al(N * 4)
    st = np.frombuffer(ws[off: off + N * 64].tobytes(), dtype=np.int64).reshape(N, 8)
    sec = st[:, :4].astype(float).mean(0) / T
    print("%-26s cycles/step: gen+probe %.0f | merge+keys+max %.0f | select %.0f | shift+write+table %.0f | staging %.0f | total %.0f | mean beam %.1f" % (
        name, sec[0], sec[1], sec[2], sec[3], st[:, 5].mean() / T, sec.sum() + st[:, 5].mean() / T, st[:, 4].mean() / T))
