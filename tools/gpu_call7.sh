#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_basecall.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r2_pytest7.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r2_pytest7.log | tail
timeout 300 python bench.py --no-cpu-baseline --no-h2d-leg > gpurun_out/r2_bench_hac_d.json 2> gpurun_out/r2_bench_hac_d.err; tail -1 gpurun_out/r2_bench_hac_d.err; python -c "
import json;d=json.load(open('gpurun_out/r2_bench_hac_d.json'));print(d['ms_per_step'],d['kernel_ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --no-h2d-leg --set beam_fuse=0 > gpurun_out/r2_bench_hac_d0.json 2> gpurun_out/r2_bench_hac_d0.err; python -c "
import json;d=json.load(open('gpurun_out/r2_bench_hac_d0.json'));print('unfused',d['ms_per_step'],d['kernel_ms_per_step'])"
timeout 300 python bench.py --quantize --no-cpu-baseline --no-h2d-leg --lanes 2 --set lstm_q8_variant=2 > gpurun_out/r2_bench_hac_q8_l2d.json 2> /dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2_bench_hac_q8_l2d.json'));print('q8 l2',d['ms_per_step'],d['kernel_ms_per_step'])"
for m in fast sup_lstm; do timeout 300 python bench.py --model $m --no-cpu-baseline --no-h2d-leg > gpurun_out/r2_bench_${m}_d.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2_bench_${m}_d.json'));print('$m',d['ms_per_step'],d['kernel_ms_per_step'])"; done
