#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python tools/lstm_q8_stats.py 0 > gpurun_out/r2_q8_stats_v0.log 2>&1; tail -7 gpurun_out/r2_q8_stats_v0.log
timeout 900 python -m pytest tests/test_gpu_q8.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r2_pytest4.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r2_pytest4.log | tail
timeout 300 python bench.py --quantize --no-cpu-baseline > gpurun_out/r2_bench_hac_q8.json 2> gpurun_out/r2_bench_hac_q8.err; tail -2 gpurun_out/r2_bench_hac_q8.err; cut -c1-330 gpurun_out/r2_bench_hac_q8.json; echo
timeout 300 python bench.py --quantize --no-cpu-baseline --lanes 2 --set lstm_q8_variant=2 > gpurun_out/r2_bench_hac_q8_l2.json 2> gpurun_out/r2_bench_hac_q8_l2.err; tail -2 gpurun_out/r2_bench_hac_q8_l2.err; cut -c1-330 gpurun_out/r2_bench_hac_q8_l2.json; echo
timeout 300 python bench.py --quantize --no-cpu-baseline --lanes 2 > gpurun_out/r2_bench_hac_q8_l2v0.json 2> gpurun_out/r2_bench_hac_q8_l2v0.err; tail -2 gpurun_out/r2_bench_hac_q8_l2v0.err; cut -c1-330 gpurun_out/r2_bench_hac_q8_l2v0.json; echo
