#!/bin/bash
# round 2, call 19: tests of the paired kernel, bench at batch 512 / 1024, end to end at batchsize 512 / 1024
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "paired or xcds or full_size" > gpurun_out/r2_pytest12.log 2>&1; tail -n 4 gpurun_out/r2_pytest12.log
timeout 300 python bench.py > gpurun_out/r2_c_hac512.json 2> gpurun_out/r2_c_hac512.err; tail -n1 gpurun_out/r2_c_hac512.json | cut -c1-330
timeout 300 python bench.py --batch 1024 > gpurun_out/r2_c_hac1024.json 2> gpurun_out/r2_c_hac1024.err; tail -n1 gpurun_out/r2_c_hac1024.json | cut -c1-330
for b in 512 1024; do timeout 280 python tools/e2e_basecall.py --reads 2000 --reps 2 --batchsize $b 2>&1 | tail -2; done > gpurun_out/r2_e2e_batch.log 2>&1; cat gpurun_out/r2_e2e_batch.log
