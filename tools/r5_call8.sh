#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
F="--model fast --steps 384 --warmup 48 --no-cpu-baseline --no-side-legs --no-h2d-leg"
timeout 300 python bench.py $F 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('fast', j['ms_per_step'], j['kernel_ms_per_step'])"
timeout 300 python bench.py $F 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('fast', j['ms_per_step'], j['kernel_ms_per_step'])"
timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python tools/decode_bench.py 2048 1667 256 2>&1 | tail -1
