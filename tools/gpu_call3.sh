#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/q8_diag.py 40 > gpurun_out/r2_q8_diag.log 2>&1; tail -12 gpurun_out/r2_q8_diag.log
for v in 0 1; do timeout 200 python tools/lstm_q8_stats.py $v > gpurun_out/r2_q8_stats_v$v.log 2>&1; tail -8 gpurun_out/r2_q8_stats_v$v.log; done
timeout 900 python -m pytest tests/test_gpu_q8.py tests/test_gpu_basecall.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r2_pytest3.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r2_pytest3.log | tail
