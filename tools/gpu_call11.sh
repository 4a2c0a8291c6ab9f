#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_basecall.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r2_pytest11.log; grep -E "passed|failed|FAILED|Error|assert" gpurun_out/r2_pytest11.log | tail
run() { tag=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-h2d-leg "$@" > gpurun_out/r2_w_$tag.json 2>gpurun_out/r2_w_$tag.err; python -c "
import json;d=json.load(open('gpurun_out/r2_w_$tag.json'));print('$tag',round(d['ms_per_step'],2),d['kernel_ms_per_step'])" || tail -3 gpurun_out/r2_w_$tag.err; }
run hac
run hac_ck0 --set beam_ckpt=0
run hac2
run hac_ck0b --set beam_ckpt=0
run q8l2 --quantize --lanes 2 --set lstm_q8_variant=2
run q8l2_ck0 --quantize --lanes 2 --set lstm_q8_variant=2 --set beam_ckpt=0
