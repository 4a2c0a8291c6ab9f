#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_q8.py -m gpu -q 2>&1 | tail -15 > gpurun_out/r2_pytest10.log; grep -E "passed|failed|FAILED|Error" gpurun_out/r2_pytest10.log | tail
run() { tag=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-h2d-leg "$@" > gpurun_out/r2_z_$tag.json 2>gpurun_out/r2_z_$tag.err; python -c "
import json;d=json.load(open('gpurun_out/r2_z_$tag.json'));print('$tag',round(d['ms_per_step'],2),d['kernel_ms_per_step'], d['roofline']['kernel'][:60])" || tail -3 gpurun_out/r2_z_$tag.err; }
run suplstm --model sup_lstm --steps 20
run suplstm_ex0 --model sup_lstm --steps 20 --set enc:lstm_exchange=0
run hac
