#!/bin/bash
# round 2, call 20: full GPU suite, profile round r02c of the default bench (two batches per engine call), bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest13.log 2>&1; tail -n 3 gpurun_out/r2_pytest13.log
PS_ARGS="--batch 1024" timeout 600 bash tools/prof_round.sh r02c > gpurun_out/r2_prof_r02c.log 2>&1; tail -n 30 gpurun_out/r2_prof_r02c.log | cut -c1-220
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py > gpurun_out/r2_d_hac.json 2> gpurun_out/r2_d_hac.err; tail -n1 gpurun_out/r2_d_hac.json | cut -c1-2000
timeout 300 python bench.py --per-call 1 > gpurun_out/r2_d_hac_pc1.json 2> gpurun_out/r2_d_hac_pc1.err; tail -n1 gpurun_out/r2_d_hac_pc1.json | cut -c1-330
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
