#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-h2d-leg "$@" > gpurun_out/r2_y_$tag.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2_y_$tag.json'));print('$tag',round(d['ms_per_step'],2),d['kernel_ms_per_step'])"; }
run base
run nt --set decode_nt=1
run base2
run nt2 --set decode_nt=1
run q8 --quantize
run q8l2 --quantize --lanes 2 --set lstm_q8_variant=2
run q8l2nt --quantize --lanes 2 --set lstm_q8_variant=2 --set decode_nt=1
timeout 600 python -m pytest tests/test_gpu_q8.py tests/test_gpu_decode.py -m gpu -q 2>&1 | tail -3
