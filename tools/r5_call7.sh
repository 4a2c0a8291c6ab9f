#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
F="--model fast --steps 384 --warmup 48 --no-cpu-baseline --no-side-legs --no-h2d-leg"
for i in 1 2; do
timeout 300 python bench.py $F 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('fast btb1', j['ms_per_step'], j['kernel_ms_per_step'])"
BONITO_HIP_LIB=$PWD/bonito_amd/libbonito_hip_expt.so timeout 300 python bench.py $F 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('fast btb2', j['ms_per_step'], j['kernel_ms_per_step'])"
done
timeout 300 python bench.py $F --set beam_fuse=0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('fast nofuse', j['ms_per_step'], j['kernel_ms_per_step'])"
timeout 300 python bench.py $F --set beam_fork=0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('fast fork0', j['ms_per_step'], j['kernel_ms_per_step'])"
