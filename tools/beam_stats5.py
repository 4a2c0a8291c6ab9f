#!/usr/bin/env python3
import os, sys
os.environ["BH_BEAM_DEBUG"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bonito_amd import decode
N, T, S = 256, 1000, 1024
C = 4 * S
g = torch.Generator(device="cuda").manual_seed(1)
sc = (torch.randn(N, T, C, generator=g, device="cuda") * 2.5).clamp(-5, 5).half()
dec = decode.CRFDecoder(N, T, C, "cuda:0", mode="beam")
dec.submit(sc).result()
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): dec.submit(sc).result()
torch.cuda.synchronize(); print("decode call %.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
ws = dec.ws.cpu().numpy()
al = lambda x: (x + 255) // 256 * 256
off = al(N * (T + 1) * S * 4) + al(N * (T + 1) * 8) + al(N * 8) + al(N * T * 4 * 4) + al(N * T * 32) + al(N * 4)
st = np.frombuffer(ws[off: off + N * 64].tobytes(), dtype=np.int64).reshape(N, 8)
sec = st[:, :4].astype(float).mean(0) / T
print("S=1024 cycles/step: gen+probe %.0f | merge+keys+max %.0f | select %.0f | shift+write+table %.0f | staging %.0f | total %.0f | mean beam %.1f" % (
    sec[0], sec[1], sec[2], sec[3], st[:, 5].mean() / T, sec.sum() + st[:, 5].mean() / T, st[:, 4].mean() / T))
