#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
F="--no-cpu-baseline --no-side-legs --no-h2d-leg"
timeout 300 python tools/decode_bench.py 512 2000 4096 2>&1 | tail -1
timeout 300 python tools/decode_bench.py 256 2000 4096 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python bench.py $F --model sup --steps 12 --warmup 4 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('sup', j['ms_per_step'], j['kernel_ms_per_step'], j['batches_per_engine_call'])"
timeout 300 python bench.py $F --model sup_lstm --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('sup_lstm', j['ms_per_step'], j['kernel_ms_per_step'])"
timeout 300 python bench.py $F --model sup --chunk 20000 --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('sup_20000', j['ms_per_step'], j['kernel_ms_per_step'])"
