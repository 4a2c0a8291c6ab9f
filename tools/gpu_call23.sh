#!/bin/bash
# round 2, call 23 (final state): full GPU suite, smoke, profile round r02e of the default bench, its JSON line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest15.log 2>&1; tail -n 2 gpurun_out/r2_pytest15.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
PS_ARGS="--batch 1024" timeout 700 bash tools/prof_round.sh r02e > gpurun_out/r2_prof_r02e.log 2>&1; sed -n 2,10p gpurun_out/r2_prof_r02e.log | cut -c1-160
cd $GRAFT_REPO_ROOT
timeout 400 python bench.py > gpurun_out/r2_j_hac.json 2> gpurun_out/r2_j_hac.err; tail -n1 gpurun_out/r2_j_hac.json | cut -c1-1900
