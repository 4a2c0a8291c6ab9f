#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
F="--no-cpu-baseline --no-side-legs --no-h2d-leg"
for pc in 1 2; do
timeout 300 python bench.py $F --model sup --steps 12 --warmup 4 --per-call $pc 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('sup per_call $pc', j['ms_per_step'], j['kernel_ms_per_step'])"
timeout 300 python bench.py $F --model sup_lstm --steps 8 --warmup 2 --per-call $pc 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('sup_lstm per_call $pc', j['ms_per_step'], j['kernel_ms_per_step'])"
done
timeout 300 python tools/decode_bench.py 512 2000 4096 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_basecall.py -m gpu -q -x 2>&1 | tail -3
