#!/usr/bin/env python3
"""Host-side wait latency probe: device time vs host time of submit()->result() under different wait strategies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bonito_amd import decode
N, T, C = 512, 1667, 256
scores = (torch.randn(N, T, C, device="cuda") * 2.5).clamp(-5, 5).half()
dec = decode.CRFDecoder(N, T, C, "cuda:0", mode="beam")
def run(strategy, reps=40):
    hs = []
    for rep in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tk = dec.submit(scores)
        t1 = time.perf_counter()
        if strategy == "sync":
            dec.done.synchronize()
        elif strategy == "query":
            while not dec.done.query():
                time.sleep(0.0002)
        elif strategy == "spin":
            while not dec.done.query():
                pass
        elif strategy == "stream":
            torch.cuda.current_stream().synchronize()
        t2 = time.perf_counter()
        r = tk.result()
        t3 = time.perf_counter()
        hs.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    tot = sorted(sum(h) for h in hs)
    worst = max(hs, key=sum)
    print("%-7s total ms: min %.2f median %.2f p90 %.2f max %.2f | worst split submit %.2f wait %.2f result %.2f | n>15ms %d" % (
        strategy, tot[0], tot[len(tot) // 2], tot[int(len(tot) * 0.9)], tot[-1], worst[0], worst[1], worst[2], sum(t > 15 for t in tot)))
def cg(path):
    try:
        return open(path).read().strip()
    except Exception as e:
        return "n/a"
print("cpu_count %s affinity %d torch threads %d cpu.max %s cfs_quota %s" % (os.cpu_count(), len(os.sched_getaffinity(0)), torch.get_num_threads(),
      cg("/sys/fs/cgroup/cpu.max"), cg("/sys/fs/cgroup/cpu/cpu.cfs_quota_us")))
if len(sys.argv) > 1:
    torch.set_num_threads(int(sys.argv[1]))
    print("torch threads now", torch.get_num_threads())
for s in ("sync", "query", "sync"):
    run(s)
