#!/usr/bin/env python3
"""Cycle statistics of lstm_layer_wgx2_kernel (two rings per workgroup), lstm_tune bit 2: where does a ring step go?
usage: lstm_stats2.py [batch=1024]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bonito_amd import synthetic, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = synthetic.make_model("hac")
model.use_koi(batchsize=N, chunksize=10000, quantize=False)
model = model.half().cuda()
sig = torch.randn(N, 1, 10000, device="cuda").half()
model(sig)
enc = model._hip
print(enc.describe().splitlines()[4])
# the product instance (no stamps): HIP-event time of the recurrent launches
enc.profile(True)
for _ in range(3):
    model(sig)
torch.cuda.synchronize()
prof = enc.profile_read()
enc.profile(False)
ms, spans = prof["lstm_rec"]
launches = spans * max(1, -(-(N // 16) // 64))
print("recurrent kernel without stamps: %.3f ms per launch (%d launches) = %.0f cycles per ring step at 2.4 GHz" % (ms / launches, launches, ms / launches * 1e-3 * 2.4e9 / 1667 / 2))
rings, nsl, T = N // 16, 32, 1667
pairs = (rings + 1) // 2
off = (rings * nsl * 4 + 64 + 7) & ~7


def stamps(tune):
    enc.set_option("lstm_tune", tune)
    model(sig); torch.cuda.synchronize(); enc.check()
    st = np.zeros((rings, nsl, 16), np.int64)
    _lib.check(_lib.lib().bh_encoder_debug_read(enc._handle, st.ctypes.data_as(C.c_void_p), st.nbytes, off))
    return st[:pairs].astype(float)                # a workgroup reports under its first ring


# the product path (main loop unrolled over four steps, no stamps inside it): total cycles, clock, re-polls
st = stamps(4)
tot = st[..., 0]
print("cycles per pair step (two ring steps): mean %.0f (min %.0f max %.0f) -> %.0f per ring step" % (tot.mean() / T, tot.min() / T, tot.max() / T, tot.mean() / T / 2))
print("validations that needed a re-poll: %.2f %% of the ring steps" % (100 * st[..., 2].mean() / T / 2))
print("shader clock during the kernel: %.2f GHz" % (tot.mean() / st[..., 13].mean() * 0.1))
# lstm_tune bit 6: the generic section code for every step (what the first four and the last steps run), with stamps per section
st = stamps(4 | 64)
tot = st[..., 0]
print("generic section code throughout (lstm_tune bit 6): %.0f cycles per ring step, clock %.2f GHz" % (tot.mean() / T / 2, tot.mean() / st[..., 13].mean() * 0.1))
names = ["barrier", "the stream (MFMAs + gates + polls + x-stream DMA + validation + LDS transpose + stores)", "re-poll rounds + bookkeeping"]
idx = [5, 10, 12]
for n, i in zip(names, idx):
    print("  %-84s %7.0f per ring step" % (n, st[..., i].mean() / T / 2))
enc.set_option("lstm_tune", 0)
