#!/bin/bash
# kernel trace only of the decode stage: bash tools/prof_decode_trace.sh <tag> N T C
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o dec -- python $R/tools/decode_bench.py "$@" > $OUT/decode_bench.log 2>&1
python $R/tools/rocprof_summary.py $(find $OUT/trace -name "*.db" | head -1) $OUT/decode_kernel_stats_$1x$2x$3.csv | grep -v "at::native" | head -6
rm -rf $OUT/trace
