#!/bin/bash
# per-kernel rocprofv3 summary of a few bench-like steps: bash tools/prof_model.sh <model> [decoder]
M=${1:-sup}; D=${2:-beam}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/pm_$M -o pm -- python $R/tools/profile_step.py --model $M --decoder $D --steps 2 > $R/gpurun_out/pm_$M.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/pm_$M -name "*.db" | head -1) $R/gpurun_out/pm_$M.csv | head -16
rm -rf $R/gpurun_out/pm_$M
