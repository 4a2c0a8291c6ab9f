#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 env "${ENVV[@]}" python bench.py --no-cpu-baseline --no-h2d-leg "$@" > gpurun_out/r2_x_$tag.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2_x_$tag.json'));print('$tag',round(d['ms_per_step'],2),d['kernel_ms_per_step'])"; }
ENVV=(A=1); run base
ENVV=(A=1); run setprio --set enc:lstm_tune=16
ENVV=(BENCH_ENC_PRIORITY=-1); run streamprio
ENVV=(BENCH_ENC_PRIORITY=-1); run both --set enc:lstm_tune=16
ENVV=(A=1); run nosleep --set enc:lstm_tune=1
bash tools/prof_round.sh r02b
