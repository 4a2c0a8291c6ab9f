#!/bin/bash
# first GPU call of the round: full GPU test suite, default bench, 2-rank bench (shared device), e2e tool
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r2_pytest1.log
tail -5 gpurun_out/r2_pytest1.log
timeout 300 python bench.py > gpurun_out/r2_bench_hac.json 2> gpurun_out/r2_bench_hac.err; tail -3 gpurun_out/r2_bench_hac.err; cat gpurun_out/r2_bench_hac.json
timeout 300 python bench.py --gpus 2 --model fast --lanes 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench_2rank.json 2> gpurun_out/r2_bench_2rank.err; tail -3 gpurun_out/r2_bench_2rank.err; cat gpurun_out/r2_bench_2rank.json
timeout 300 python tools/e2e_basecall.py --model fast --reads 400 --devices 0,0 --reps 1 2>&1 | tail -3 > gpurun_out/r2_e2e_2rank.log; cat gpurun_out/r2_e2e_2rank.log
