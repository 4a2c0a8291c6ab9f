#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -120 > gpurun_out/r2_pytest2.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r2_pytest2.log | tail -30
for v in 0 1; do timeout 200 python tools/lstm_q8_stats.py $v > gpurun_out/r2_q8_stats_v$v.log 2>&1; tail -12 gpurun_out/r2_q8_stats_v$v.log; done
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_hac_b.json 2> gpurun_out/r2_bench_hac_b.err; tail -2 gpurun_out/r2_bench_hac_b.err; cut -c1-700 gpurun_out/r2_bench_hac_b.json
timeout 300 python bench.py --quantize --no-cpu-baseline > gpurun_out/r2_bench_hac_q8.json 2> gpurun_out/r2_bench_hac_q8.err; tail -2 gpurun_out/r2_bench_hac_q8.err; cut -c1-700 gpurun_out/r2_bench_hac_q8.json
timeout 300 python bench.py --quantize --no-cpu-baseline --set lstm_q8_variant=1 > gpurun_out/r2_bench_hac_q8v1.json 2> gpurun_out/r2_bench_hac_q8v1.err; tail -2 gpurun_out/r2_bench_hac_q8v1.err; cut -c1-700 gpurun_out/r2_bench_hac_q8v1.json
ls gpurun_out/parity_* 2>/dev/null | head -30
