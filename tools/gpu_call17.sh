#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 200 python - > gpurun_out/r2_pair_stats.log 2>&1 <<'PY'
import ctypes as C, numpy as np, torch
from bonito_amd import synthetic, _lib
from bonito_amd.engine import HipEncoder
m = synthetic.make_model("hac", batchsize=1024, chunksize=10000)
x = torch.randn(1024, 1, 10000, device="cuda").half()
enc = HipEncoder(m.encoder, batchsize=1024, chunksize=10000)
enc(x); enc.set_option("lstm_tune", 4); enc(x); torch.cuda.synchronize(); enc.check()
rings, nsl, T = 64, 32, 1667
off = (rings * nsl * 4 + 64 + 7) & ~7
st = np.zeros((rings, nsl, 16), np.int64)
_lib.check(_lib.lib().bh_encoder_debug_read(enc._handle, st.ctypes.data_as(C.c_void_p), st.nbytes, off))
st = st[:32]
print("cycles per double step %.0f (per ring step %.0f); poll check %.0f, wait + barrier %.0f per double step; clock %.2f GHz" % (
    st[..., 0].mean() / T, st[..., 0].mean() / T / 2, st[..., 1].mean() / T, st[..., 5].mean() / T, st[..., 0].mean() / st[..., 13].mean() * 0.1))
print("per ring step: x DMA issue + recurrent MFMAs %.0f | gates woven with the input projection %.0f | read-back + transpose + stores %.0f | test + polls %.0f" % tuple(
    st[..., i].mean() / T / 2 for i in (10, 11, 12, 8)))
print("re-poll rounds per ring step: %.4f" % (st[..., 9].mean() / T / 2))
PY
cat gpurun_out/r2_pair_stats.log | tail -3
