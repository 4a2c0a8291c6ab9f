#!/usr/bin/env python3
"""Board power and shader clock (rocm-smi) while one kernel class runs back to back: is it at the power limit?
usage (GPU box): python tools/power_probe.py            -> one line per workload: idle, persistent GEMM (K = 2048 and 512), paired recurrent kernel"""
import os, subprocess, sys, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bonito_amd import _lib, synthetic
INF = float("inf")
dev = torch.device("cuda", 0)


def sample():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
    keep = [ln.strip() for ln in out.splitlines() if re.search(r"Power|sclk|Max Graphics", ln)]
    return " | ".join(re.sub(r"\s+", " ", ln) for ln in keep)


def probe(name, fn, seconds=5.0):
    stop = [False]
    def loop():
        while not stop[0]:
            fn()
            torch.cuda.synchronize()
    th = threading.Thread(target=loop); th.start()
    time.sleep(seconds * 0.5)
    a = sample(); time.sleep(seconds * 0.25); b = sample()
    stop[0] = True; th.join()
    print("%-44s %s\n%-44s %s" % (name, a, "", b), flush=True)


print("%-44s %s" % ("idle", sample()), flush=True)
lib = _lib.lib()
for M, N, K in ((256000, 512, 2048), (256000, 1536, 512)):
    x = (torch.randn(M, K, device=dev) * 0.5).half(); w = (torch.randn(N, K, device=dev) * 0.2).half()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    def gemm():
        for _ in range(50):
            _lib.check(lib.bh_linear(_lib.ptr(x), _lib.ptr(w), None, _lib.ptr(out), M, N, K, K, K, N, 0, 1.0, -INF, INF, 0, 0, 0, 0, 0, _lib.stream_ptr()), "bh_linear")
    probe("gemm_big_kernel %d x %d x %d" % (M, N, K), gemm)
    full = torch.empty((M, N), dtype=torch.float16, device=dev)
    def vendor():
        for _ in range(50):
            torch.matmul(x, w.t(), out=full)
    probe("torch.matmul (vendor) %d x %d x %d" % (M, N, K), vendor)
    del x, w, out, full
model = synthetic.make_model("hac")
model.use_koi(batchsize=1024, chunksize=10000, quantize=False)
model = model.half().cuda()
sig = torch.randn(1024, 1, 10000, device="cuda").half()
probe("hac encoder 1024 x 10000 (92 % recurrent kernel)", lambda: model(sig), 6.0)
