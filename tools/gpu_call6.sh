#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r2_pytest6.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r2_pytest6.log | tail -20
for m in fast sup sup_lstm; do timeout 400 python bench.py --model $m --no-cpu-baseline > gpurun_out/r2_bench_$m.json 2> gpurun_out/r2_bench_$m.err; tail -1 gpurun_out/r2_bench_$m.err; cut -c1-300 gpurun_out/r2_bench_$m.json; echo; done
timeout 300 python bench.py --model fast --quantize --no-cpu-baseline > gpurun_out/r2_bench_fast_q8.json 2> gpurun_out/r2_bench_fast_q8.err; tail -1 gpurun_out/r2_bench_fast_q8.err; cut -c1-300 gpurun_out/r2_bench_fast_q8.json; echo
timeout 300 python tools/e2e_basecall.py --model hac --reads 1500 --reps 2 2>&1 | tail -2
