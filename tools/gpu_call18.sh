#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python - > gpurun_out/r2_pair_diff.log 2>&1 <<'PY'
import torch
from bonito_amd import synthetic
from bonito_amd.engine import HipEncoder
m = synthetic.make_model("hac", batchsize=1024, chunksize=600)
# only the first recurrent layer: truncate the encoder? compare final scores and count mismatching elements
x = torch.randn(1024, 1, 600, generator=torch.Generator().manual_seed(3)).half().cuda()
outs = []
for pair in (0, 1):
    enc = HipEncoder(m.encoder, batchsize=1024, chunksize=600); enc.set_option("lstm_pair", pair)
    outs.append(enc(x).float().clone()); enc.check(); enc.close()
d = (outs[0] - outs[1]).abs()
print("max abs diff %.3e, mean %.3e, fraction of differing elements %.4f, nan %d" % (d.max().item(), d.mean().item(), (d > 0).float().mean().item(), int(torch.isnan(outs[1]).sum())))
bad = (d > 0).nonzero()
print("first differing indices (n, t, c):", bad[:5].tolist())
print("differing chunks: %d of 1024; first t with a difference: %d" % (len(torch.unique(bad[:, 0])), int(bad[:, 1].min())))
ch = torch.unique(bad[:, 0]); print("chunk ids mod 16 histogram:", torch.bincount(ch % 16, minlength=16).tolist()); print("chunk id // 16 (rings) sample:", torch.unique(ch // 16)[:20].tolist())
PY
cat gpurun_out/r2_pair_diff.log | tail -5
