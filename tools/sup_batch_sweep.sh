#!/bin/bash
# does the v5 transformer run faster per chunk when a call's activations fit the 256 MB Infinity Cache? (kernel classes per chunk at several call sizes)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for B in 32 64 128 256 512; do
  python bench.py --model sup --batch $B --per-call 1 --steps 6 --warmup 2 --no-cpu-baseline --parity-chunks 0 --no-h2d-leg --no-side-legs --repeats 1 --warmup-seconds 0.5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=$B
k=j['kernel_ms_per_step']
print('batch %4d: %.3f ms per chunk; per chunk (us):' % (b, j['ms_per_step']/b), {n: round(1e3*v/b,1) for n,v in k.items()})
"
done
