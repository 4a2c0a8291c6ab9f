#!/bin/bash
# round 2, call 16: two rings per workgroup (lstm_layer_wgx2_kernel): same bytes as two launches? how fast?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 500 python - > gpurun_out/r2_pair_check.log 2>&1 <<'PY'
import torch, time
from bonito_amd import synthetic, nn as bnn
from bonito_amd.engine import HipEncoder
def run(model, x, pair, tune=0):
    enc = HipEncoder(model, batchsize=x.shape[0], chunksize=x.shape[-1])
    enc.set_option("lstm_pair", pair); enc.set_option("lstm_tune", tune)
    y = enc(x).clone(); y2 = enc(x).clone()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): enc(x)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    enc.check(); enc.close()
    assert torch.equal(y, y2)
    return y, dt
m = synthetic.make_model("hac", batchsize=1024, chunksize=2400)
for N in (1024, 528, 1008, 640):
    x = torch.randn(N, 1, 2400, generator=torch.Generator().manual_seed(3)).half().cuda()
    a, ta = run(m.encoder, x, 0); b, tb = run(m.encoder, x, 1); c, tc = run(m.encoder, x, 1, 32)
    print("hac %dx2400: paired == two launches: %s, across XCDs: %s   (%.2f ms vs %.2f ms)" % (N, torch.equal(a, b), torch.equal(a, c), tb * 1e3, ta * 1e3), flush=True)
for H, sl in [(256, 3), (192, 3), (288, 3)]:
    torch.manual_seed(H)
    model = bnn.from_dict(synthetic.lstm_crf_encoder_config(H, sl, n_lstm=3))
    synthetic.randomise_batchnorm_(model)
    for N in (1024, 1500):
        xx = torch.randn(N, 1, 900).half().cuda()
        a, ta = run(model, xx, 0); b, tb = run(model, xx, 1)
        print(H, N, "equal:", torch.equal(a, b), "(%.2f vs %.2f ms)" % (tb * 1e3, ta * 1e3), flush=True)
m = synthetic.make_model("hac", batchsize=1024, chunksize=10000)
x = torch.randn(1024, 1, 10000, generator=torch.Generator().manual_seed(25)).half().cuda()
a, ta = run(m.encoder, x, 0); b, tb = run(m.encoder, x, 1)
print("hac 1024x10000 equal: %s; encoder %.2f ms paired vs %.2f ms as two launches per layer" % (torch.equal(a, b), tb * 1e3, ta * 1e3))
PY
cat gpurun_out/r2_pair_check.log | tail -14; bash tools/gpu_call17.sh
timeout 300 python bench.py --steps 30 --warmup 5 --batch 1024 > gpurun_out/r2_b_pair.json 2> gpurun_out/r2_b_pair.err
timeout 300 python bench.py --steps 30 --warmup 5 --batch 1024 --set enc:lstm_pair=0 > gpurun_out/r2_b_pair0.json 2> gpurun_out/r2_b_pair0.err
tail -n1 gpurun_out/r2_b_pair.json | cut -c1-1700; tail -n1 gpurun_out/r2_b_pair0.json | cut -c1-400; tail -n 3 gpurun_out/r2_b_pair.err
