#!/bin/bash
# A/B of the four-wave GEMM's MFMA shape inside the engine: bench.py legs with gemm_tile16 = 0 / 1, same box, alternating.
# usage: bash tools/t16_ab.sh <tag>   -> gpurun_out/<tag>_t16_ab.log
tag=${1:-ab}
out=gpurun_out/${tag}_t16_ab.log
mkdir -p gpurun_out; : > $out
for rep in 1 2; do
  for model in sup hac sup_lstm; do
    for t in 0 1; do
      echo "== $model gemm_tile16=$t pass $rep" >> $out
      python bench.py --model $model --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs --no-h2d-leg --parity-chunks 0 --set gemm_tile16=$t 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d.get('regions_ms_per_step'))" >> $out
    done
  done
done
