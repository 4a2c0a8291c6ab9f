#!/bin/bash
# builds and runs tools/mfma_tile_energy.hip on the GPU box with board power / clock sampled beside it (rocm-smi, twice a second)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_tile_energy tools/mfma_tile_energy.hip || exit 1
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr -s ' ' | tr '\n' '|'; echo; sleep 0.5; done ) > gpurun_out/_power.log &
SMI=$!
/tmp/mfma_tile_energy
kill $SMI
echo "-- board power / shader clock samples while the variants ran (rocm-smi, every 0.5 s; one line per sample, first and every fourth shown)"
awk 'NR==1 || NR%4==0' gpurun_out/_power.log | cut -c1-200 | head -40
rm -f gpurun_out/_power.log
