#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r2_pytest5.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r2_pytest5.log | tail
for ex in 1 0; do timeout 200 python tools/lstm_stats.py 2 $ex > gpurun_out/r2_lstm_stats_ex$ex.log 2>&1; tail -9 gpurun_out/r2_lstm_stats_ex$ex.log; done
timeout 200 python tools/lstm_q8_stats.py 0 > gpurun_out/r2_q8_stats_v0.log 2>&1; tail -6 gpurun_out/r2_q8_stats_v0.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_hac_c.json 2> gpurun_out/r2_bench_hac_c.err; tail -2 gpurun_out/r2_bench_hac_c.err; cut -c1-330 gpurun_out/r2_bench_hac_c.json; echo
timeout 300 python bench.py --no-cpu-baseline --lanes 2 > gpurun_out/r2_bench_hac_l2.json 2> gpurun_out/r2_bench_hac_l2.err; tail -2 gpurun_out/r2_bench_hac_l2.err; cut -c1-330 gpurun_out/r2_bench_hac_l2.json; echo
timeout 300 python bench.py --quantize --no-cpu-baseline --lanes 2 --set lstm_q8_variant=2 > gpurun_out/r2_bench_hac_q8_l2.json 2> gpurun_out/r2_bench_hac_q8_l2.err; tail -2 gpurun_out/r2_bench_hac_q8_l2.err; cut -c1-330 gpurun_out/r2_bench_hac_q8_l2.json; echo
