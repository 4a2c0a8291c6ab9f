#!/bin/bash
# Runs the tile16 laboratory binaries (tools/gemm_lab.py base t16_*) on the GPU box: per variant and shape time, TFLOP/s, cycles per K-tile and
# per epilogue of workgroups 0 / 133, shader clock. usage: bash tools/gemm_lab_t16.sh <tag>   -> gpurun_out/<tag>_gemm_lab_t16.log
tag=${1:-lab}
out=gpurun_out/${tag}_gemm_lab_t16.log
mkdir -p gpurun_out
: > $out
for rep in 1 2; do
  echo "== base (32x32x16 stream), pass $rep" >> $out
  LAB_REPS=${LAB_REPS:-40} LAB_T16=0 timeout 120 build/gemmlab/lab_base >> $out 2>&1
  for v in build/gemmlab/lab_t16_*; do
    echo "== $(basename $v), pass $rep" >> $out
    LAB_REPS=${LAB_REPS:-40} LAB_T16=1 timeout 120 $v >> $out 2>&1
  done
done
