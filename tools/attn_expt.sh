#!/bin/bash
# attention kernel A/B + elimination experiments (tools/attn_bench.py): the product library (version 1 = rounds 2-5, version 2 = round 6) and the
# experiment library (BH_EXTRA_ATTN_FLAGS=-DBH_ATTN_EXPT python build.py): bit 0 no exponentials, 1 no PV MFMAs, 2 no QK^T MFMAs, 3 no staging,
# 4 no barriers; correct-result placement variants: 32 = next block's loads requested behind the QK^T phase, 64 = V task on the coalesced mapping
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== version 1 (rounds 2-5)"; python tools/attn_bench.py --set attn_version=1 2>&1 | tail -9
echo "== version 2 (round 6)";   python tools/attn_bench.py --set attn_version=2 2>&1 | tail -9
echo "== version 2, 8 waves";   python tools/attn_bench.py --set attn_version=2 --set attn_waves=8 2>&1 | grep time
if [ -f bonito_amd/libbonito_hip_expt.so ]; then
  for E in ${EXPTS:-1 2 4 7 8 16 24 31 32 64 96}; do
    echo "== version 2, experiment bits $E (wrong results on purpose)"
    BONITO_HIP_LIB=$PWD/bonito_amd/libbonito_hip_expt.so python tools/attn_bench.py --set attn_expt=$E 2>&1 | grep time
  done
fi
