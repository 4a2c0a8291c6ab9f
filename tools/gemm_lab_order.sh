#!/bin/bash
# Work order of gemm_w4_kernel inside an XCD's share (LAB_ORDER 0: token blocks fastest, 1: feature groups fastest) x feature tiles per block
# (LAB_GF), on the final 16x16x32 stream. usage: bash tools/gemm_lab_order.sh <tag>  -> gpurun_out/<tag>_gemm_lab_order.log
tag=${1:-lab}
out=gpurun_out/${tag}_gemm_lab_order.log
mkdir -p gpurun_out; : > $out
for rep in 1 2; do
  for order in 0 1; do
    for gf in 0 2 8; do
      echo "== order${order}_gf${gf}, pass $rep" >> $out
      LAB_REPS=${LAB_REPS:-80} LAB_T16=1 LAB_ORDER=$order LAB_GF=$gf timeout 120 build/gemmlab/lab_t16_base >> $out 2>&1
    done
  done
done
