#!/usr/bin/env python3
"""Per-wave cycle statistics of the 8-bit recurrent kernel (lstm_tune bit 2): where does a time step go?
    python tools/lstm_q8_stats.py [variant]      variant 0: 12 units per wave, 1 workgroup per CU; 1: 4 units, 3 per CU"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bonito_amd import synthetic, _lib, decode
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
decode.set_option("lstm_q8_variant", variant)
model = synthetic.make_model("hac")
model.use_koi(batchsize=512, chunksize=10000, quantize=True)
model = model.half().cuda()
sig = torch.randn(512, 1, 10000, device="cuda").half()
model(sig)
enc = model._hip
print(enc.describe())
enc.profile(True)
for _ in range(3):
    model(sig)
torch.cuda.synchronize(); enc.check()
prof = enc.profile_read(); enc.profile(False)
print({k: (round(v[0] / 3, 3), v[1] // 3) for k, v in prof.items() if v[1]})
enc.set_option("lstm_tune", 4)
model(sig); torch.cuda.synchronize(); enc.check()
rings, nsl, T = 32, (96 if variant == 1 else 32), 1667
xcc = np.zeros(rings * nsl, np.int32)
_lib.check(_lib.lib().bh_encoder_debug_read(enc._handle, xcc.ctypes.data_as(C.c_void_p), xcc.nbytes, 0))
off = (rings * nsl * 4 + 64 + 7) & ~7
st = np.zeros((rings, nsl, 16), np.int64)
_lib.check(_lib.lib().bh_encoder_debug_read(enc._handle, st.ctypes.data_as(C.c_void_p), st.nbytes, off))
x = xcc.reshape(rings, nsl)
print("rings whose members share one XCD:", int((x.min(1) == x.max(1)).sum()), "of", rings)
tot, poll, rounds, first, xph, bar, rec = [st[..., i].astype(float) for i in range(7)]
print("cycles/step total mean %.0f (min %.0f max %.0f)" % (tot.mean() / T, tot.min() / T, tot.max() / T))
print("cycles/step: poll check + re-poll %.0f | barrier %.0f | recurrent + gates + store + poll issue %.0f | input projection %.0f"
      % (poll.mean() / T, bar.mean() / T, rec.mean() / T, xph.mean() / T))
print("cycles/step inside the recurrent part: LDS reads + h-MFMAs %.0f | gates + transpose %.0f" % (st[..., 7].astype(float).mean() / T, st[..., 8].astype(float).mean() / T))
print("poll rounds/step mean %.2f; first round already complete in %.1f%% of steps" % (rounds.mean() / T, 100 * first.mean() / T))
