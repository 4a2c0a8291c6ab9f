#!/bin/bash
# round 2, call 22: profile round r02d of the final default bench (4 batches per engine call, 200 steps) + its JSON line + per-call variants
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PS_ARGS="--batch 1024" timeout 700 bash tools/prof_round.sh r02d > gpurun_out/r2_prof_r02d.log 2>&1; head -n 16 gpurun_out/r2_prof_r02d.log | cut -c1-200
cd $GRAFT_REPO_ROOT
timeout 400 python bench.py > gpurun_out/r2_g_hac.json 2> gpurun_out/r2_g_hac.err; tail -n1 gpurun_out/r2_g_hac.json | cut -c1-1900
timeout 400 python bench.py --per-call 2 --no-cpu-baseline > gpurun_out/r2_g_hac_pc2.json 2> /dev/null; tail -n1 gpurun_out/r2_g_hac_pc2.json | cut -c1-300
timeout 400 python bench.py --per-call 1 --no-cpu-baseline > gpurun_out/r2_g_hac_pc1.json 2> /dev/null; tail -n1 gpurun_out/r2_g_hac_pc1.json | cut -c1-300
