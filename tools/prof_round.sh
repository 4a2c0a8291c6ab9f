#!/bin/bash
# Round profile on the GPU box: kernel trace of the exact bench command + separate PMC passes (HBM read / write bytes).
# usage: bash tools/prof_round.sh <tag> [extra bench / profile_step flags, e.g. --quantize]
#        outputs under gpurun_out/prof_<tag>/ (summaries are copied to profiles/ by hand)
#        PS_ARGS: flags for the PMC workload (tools/profile_step.py) when they differ from the bench flags, e.g. PS_ARGS="--batch 1024"
#        for the default bench (two batches of 512 per engine call)
TAG=${1:-r02}
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $R/bench.py --no-cpu-baseline --no-h2d-leg --no-side-legs "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-300
python $R/tools/rocprof_summary.py $(find $OUT/trace -name "*.db" | head -1) $OUT/bench_kernel_stats.csv | head -14
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_rd -o rd -- python $R/tools/profile_step.py --steps 1 ${PS_ARGS:-"$@"} > $OUT/pmc_rd.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_wr -o wr -- python $R/tools/profile_step.py --steps 1 ${PS_ARGS:-"$@"} > $OUT/pmc_wr.log 2>&1
python $R/tools/pmc_summary.py $(find $OUT/pmc_rd -name "*.db" | head -1) > $OUT/pmc_rd.txt 2>&1
python $R/tools/pmc_summary.py $(find $OUT/pmc_wr -name "*.db" | head -1) > $OUT/pmc_wr.txt 2>&1
grep -v "^\[" $OUT/pmc_rd.txt | head -10; grep -v "^\[" $OUT/pmc_wr.txt | head -10
rm -rf $OUT/trace $OUT/pmc_rd $OUT/pmc_wr
