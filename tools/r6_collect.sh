#!/bin/bash
# here, after tools/r6_profiles.sh ran on a GPU box: copy the per-config kernel tables into profiles/ and turn the PMC passes into
# profiles/<tag>_<cfg>_pmc_hbm_bytes.txt + the entries of profiles/pmc_traffic.json that bench.py's roofline.traffic reads
#   usage: bash tools/r6_collect.sh <run tag, e.g. r06k> [<name in profiles/, default r06>]
RUN=${1:-r06}; OUT=${2:-r06}
cd $(dirname $0)/..
declare -A WL=( [hac]="hac 1024x10000" [hac_quantize]="hac quantize 2048x10000" [fast]="fast 4096x10000" [sup]="sup 512x12000" [sup_20000]="sup 512x20000" [sup_lstm]="sup_lstm 256x20000" )
for CFG in hac hac_quantize fast sup sup_20000 sup_lstm; do
  D=gpurun_out/prof_${RUN}_$CFG
  [ -d $D ] || continue
  cp $D/bench_kernel_stats.csv profiles/${OUT}_${CFG}_kernel_stats.csv
  python tools/pmc_traffic.py $D ${OUT}_${CFG} "${WL[$CFG]}" | tail -4
done
