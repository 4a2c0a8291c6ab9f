#!/bin/bash
# Decode stage alone under rocprofv3: kernel trace (per-kernel time) + SQ counter passes of tools/decode_bench.py.
#   usage: bash tools/prof_decode.sh <tag> [N T C] [viterbi]     (FETCH_SIZE / WRITE_SIZE: KiB per dispatch, own passes)     outputs: gpurun_out/prof_<tag>/decode_{kernel_stats.csv,sq_counters.txt}
TAG=${1:-r05}
shift
ARGS=${@:-2048 1667 1024}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o dec -- python $R/tools/decode_bench.py $ARGS > $OUT/decode_bench.log 2>&1
python $R/tools/rocprof_summary.py $(find $OUT/trace -name "*.db" | head -1) $OUT/decode_kernel_stats.csv | head -8
rm -rf $OUT/trace
: > $OUT/decode_sq_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS SQ_INSTS_BRANCH" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  BH_DECODE_BENCH_REPS=1 rocprofv3 --kernel-trace --pmc $set -d $OUT/sq$i -o sq -- python $R/tools/decode_bench.py $ARGS > $OUT/sq$i.log 2>&1
  db=$(find $OUT/sq$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_summary.py $db 2>/dev/null | grep -v "at::native\|rocclr\|^\[" >> $OUT/decode_sq_counters.txt; else echo "pass $i ($set) failed: $(tail -n 2 $OUT/sq$i.log | head -1)" >> $OUT/decode_sq_counters.txt; fi
  rm -rf $OUT/sq$i
done
cat $OUT/decode_sq_counters.txt | cut -c1-150
