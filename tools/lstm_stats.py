#!/usr/bin/env python3
"""Per-wave cycle statistics of the fused LSTM kernel (lstm_tune bit 2): where does a time step go?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bonito_amd import synthetic, _lib
model = synthetic.make_model("hac")
model.use_koi(batchsize=512, chunksize=10000, quantize=False)
model = model.half().cuda()
sig = torch.randn(512, 1, 10000, device="cuda").half()
model(sig)
enc = model._hip
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2        # lstm_fused option: 2 = workgroup-shared, 1 = per-wave
enc.set_option("lstm_fused", mode)
if len(sys.argv) > 2:
    enc.set_option("lstm_exchange", int(sys.argv[2]))         # 1 (default): ring-buffer hand-off, 0: through the output tensor
print(enc.describe().splitlines()[4])
enc.set_option("lstm_tune", 4)
model(sig); torch.cuda.synchronize(); enc.check()
rings, nsl, T = 32, (32 if mode >= 2 else 24), 1667
xcc = np.zeros(rings * nsl, np.int32)
_lib.check(_lib.lib().bh_encoder_debug_read(enc._handle, xcc.ctypes.data_as(C.c_void_p), xcc.nbytes, 0))
off = (rings * nsl * 4 + 64 + 7) & ~7
st = np.zeros((rings, nsl, 16 if mode >= 2 else 8), np.int64)
_lib.check(_lib.lib().bh_encoder_debug_read(enc._handle, st.ctypes.data_as(C.c_void_p), st.nbytes, off))
x = xcc.reshape(rings, nsl)
print("rings whose members share one XCD:", int((x.min(1) == x.max(1)).sum()), "of", rings, " xcc ids ring0:", x[0][:8])
tot, poll, rounds, first = [st[..., i].astype(float) for i in range(4)]
print("cycles/step total  mean %.0f  (min %.0f max %.0f)" % (tot.mean() / T, tot.min() / T, tot.max() / T))
print("cycles/step in poll mean %.0f  -> %.0f%% of the step" % (poll.mean() / T, 100 * poll.mean() / tot.mean()))
print("poll rounds/step mean %.2f; first round already complete in %.1f%% of steps" % (rounds.mean() / T, 100 * first.mean() / T))
xph, re, rec = [st[..., i].astype(float).mean() / T for i in (4, 5, 6)]
if mode >= 2:
    print("cycles/step: input projection %.0f | workgroup barrier %.0f | recurrent + gates + store + poll issue %.0f" % (xph, re, rec))
else:
    print("cycles/step: issue polls + input projection %.0f | re-poll rounds %.0f | x fetch + recurrent + gates + store %.0f" % (xph, re, rec))
hist = st[..., 7].astype(np.uint64)
h = [((hist >> np.uint64(16 * i)) & np.uint64(0xffff)).astype(float).mean() / T for i in range(4)]
print("rounds histogram (1,2,3,>=4): %s" % " ".join("%.3f" % v for v in h))
if mode >= 2:
    names = ["x->LDS (A)", "x fetch issue (C)", "recurrent MFMAs", "gates", "x wait + transpose + store + poll issue"]
    print("cycles/step: " + " | ".join("%s %.0f" % (n, st[..., 8 + i].astype(float).mean() / T) for i, n in enumerate(names)))
if mode >= 2:
    print("shader clock during the kernel: %.2f GHz (cycle counter / 100 MHz real-time counter)" % (tot.mean() / st[..., 13].astype(float).mean() * 0.1))
print("clock: readcyclecounter ticks; 100 MHz or shader clock depending on source")
