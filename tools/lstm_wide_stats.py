#!/usr/bin/env python3
"""Per-wave cycle stamps of lstm_layer_wide_kernel (lstm_tune bit 2): where does a half-step of the 1024-wide recurrent layer go?
    python tools/lstm_wide_stats.py [chunk] [extra lstm_tune bits]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bonito_amd import synthetic, _lib
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
model = synthetic.make_model("sup_lstm", batchsize=256, chunksize=chunk)
model.use_koi(batchsize=256, chunksize=chunk, quantize=False)
model = model.half().cuda()
sig = torch.randn(256, 1, chunk, device="cuda").half()
model(sig)
enc = model._hip
if extra:
    enc.set_option("lstm_tune", extra)      # the timed passes run with the extra bits too (e.g. 128: round-4 poll order)
print([ln for ln in enc.describe().splitlines() if "lstm" in ln][0])
enc.profile(True)
for _ in range(3):
    model(sig)
torch.cuda.synchronize(); enc.check()
prof = enc.profile_read(); enc.profile(False)
print({k: (round(v[0] / 3, 3), v[1] // 3) for k, v in prof.items() if v[1]})
enc.set_option("lstm_tune", 4 | extra)
out = model(sig); torch.cuda.synchronize(); enc.check()
T = out.shape[1]
rings, nsl = 8, 128
off = (rings * nsl * 4 + 64 + 7) & ~7
st = np.zeros((rings, nsl, 16), np.int64)
_lib.check(_lib.lib().bh_encoder_debug_read(enc._handle, st.ctypes.data_as(C.c_void_p), st.nbytes, off))
tot = st[..., 0].astype(float)
f = lambda i: st[..., i].astype(float).mean() / T
print("T = %d steps; cycles/step total mean %.0f (min %.0f max %.0f); clock %.2f GHz" % (T, tot.mean() / T, tot.min() / T, tot.max() / T, tot.mean() / st[..., 6].astype(float).mean() * 0.1))
print("cycles/step (both half-steps): poll check + LDS write %.0f | barrier %.0f | MFMAs (+ other tile's polls) %.0f | gates + publish %.0f | re-poll rounds/step %.2f"
      % (f(1), f(2), f(3), f(4), f(5)))
print("  of the poll check: until the first sentinel test is decided (wait for the poll loads) %.0f" % f(7))
