#!/bin/bash
# Runs a command once per library variant under build/variants/ (lib_<name>.so): usage: bash tools/gpu_variants.sh <tag> <command...>
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
TAG=$1; shift
cp bonito_amd/libbonito_hip.so /tmp/lib_keep.so
for v in build/variants/lib_*.so; do
  n=$(basename $v .so); cp $v bonito_amd/libbonito_hip.so
  echo "=== $n"; timeout 300 "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_$n.log | tail -n 8
done
cp /tmp/lib_keep.so bonito_amd/libbonito_hip.so
