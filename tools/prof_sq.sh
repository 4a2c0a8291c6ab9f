#!/bin/bash
# SQ counters of the recurrent kernel (where do a wave's cycles go: matrix pipe busy, vector ALU issuing, waiting), one rocprofv3 --pmc pass
# each over one 1024-chunk step (tools/profile_step.py), summarised per kernel into gpurun_out/prof_<tag>/sq_counters.txt.
#   usage: bash tools/prof_sq.sh <tag> [profile_step.py arguments, default: --batch 1024 (hac)]
TAG=${1:-r03}
shift
PS=${@:---batch 1024}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counter_names.txt
: > $OUT/sq_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/sq$i -o sq -- python $R/tools/profile_step.py --steps 1 $PS > $OUT/sq$i.log 2>&1
  db=$(find $OUT/sq$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_summary.py $db 2>/dev/null | grep -v "at::native\|rocclr\|^\[" >> $OUT/sq_counters.txt; else echo "pass $i ($set) failed: $(tail -n 2 $OUT/sq$i.log | head -1)" >> $OUT/sq_counters.txt; fi
  rm -rf $OUT/sq$i
done
cat $OUT/sq_counters.txt | cut -c1-150
