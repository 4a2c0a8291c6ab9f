#!/bin/bash
# Round 6 (review item 3): rocprofv3 evidence for EVERY configuration of the bench line, on one box, from the tree as it is:
#   per config  - kernel trace + stats of the bench command itself (`bench.py --model ... --no-side-legs`)  -> <tag>_<cfg>_kernel_stats.csv
#               - two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only beside them)     -> <tag>_<cfg>_pmc_{rd,wr}.txt
# usage (on the GPU box): bash tools/r6_profiles.sh <tag> [cfg ...]      cfgs: hac hac_quantize fast sup sup_20000 sup_lstm
# afterwards, here: python tools/pmc_traffic.py gpurun_out/prof_<tag>_<cfg> <tag>_<cfg> "<workload>"   (tools/r6_collect.sh does all of it)
TAG=${1:-r06}; shift
CFGS=${@:-hac hac_quantize fast sup sup_20000 sup_lstm}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for CFG in $CFGS; do
  case $CFG in
    hac)          BENCH="--steps 40 --warmup 8";                                   STEP="--batch 1024" ;;
    hac_quantize) BENCH="--quantize --steps 48 --warmup 8";                        STEP="--quantize --batch 2048 --set lstm_q8_variant=2" ;;
    fast)         BENCH="--model fast --steps 96 --warmup 48";                     STEP="--model fast --batch 4096" ;;
    sup)          BENCH="--model sup --steps 8 --warmup 2";                        STEP="--model sup --batch 512 --chunk 12000" ;;
    sup_20000)    BENCH="--model sup --chunk 20000 --steps 4 --warmup 2";          STEP="--model sup --batch 512 --chunk 20000" ;;
    sup_lstm)     BENCH="--model sup_lstm --steps 4 --warmup 2";                   STEP="--model sup_lstm --batch 512 --chunk 20000" ;;
    *) echo "unknown config $CFG"; continue ;;
  esac
  OUT=$R/gpurun_out/prof_${TAG}_$CFG
  mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $R/bench.py $BENCH --no-cpu-baseline --parity-chunks 0 --no-h2d-leg --no-side-legs --repeats 1 --warmup-seconds 1.0 > $OUT/bench.log 2>&1
  tail -1 $OUT/bench.log | cut -c1-200
  python $R/tools/rocprof_summary.py $(find $OUT/trace -name "*.db" | head -1) $OUT/bench_kernel_stats.csv | head -8
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_rd -o rd -- python $R/tools/profile_step.py --steps 1 $STEP > $OUT/pmc_rd.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_wr -o wr -- python $R/tools/profile_step.py --steps 1 $STEP > $OUT/pmc_wr.log 2>&1
  python $R/tools/pmc_summary.py $(find $OUT/pmc_rd -name "*.db" | head -1) > $OUT/pmc_rd.txt 2>&1
  python $R/tools/pmc_summary.py $(find $OUT/pmc_wr -name "*.db" | head -1) > $OUT/pmc_wr.txt 2>&1
  grep -v "^\[" $OUT/pmc_rd.txt | head -6 | cut -c1-110
  rm -rf $OUT/trace $OUT/pmc_rd $OUT/pmc_wr
done
