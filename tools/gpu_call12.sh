#!/bin/bash
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp
OUT=$R/gpurun_out/prof_r02c; mkdir -p $OUT
cat > /tmp/ck_step.py <<PY
import sys; sys.path.insert(0, "$R")
import torch
from bonito_amd import decode, synthetic
decode.set_option("beam_ckpt", 1)
sc = (torch.randn(512, 1667, 1024, device="cuda") * 2.5).clamp(-5, 5).half()
decode.beam_search(sc); torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_rd -o rd -- python /tmp/ck_step.py > $OUT/pmc_rd.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_wr -o wr -- python /tmp/ck_step.py > $OUT/pmc_wr.log 2>&1
python $R/tools/pmc_summary.py $(find $OUT/pmc_rd -name "*.db" | head -1) > $OUT/pmc_rd.txt 2>&1
python $R/tools/pmc_summary.py $(find $OUT/pmc_wr -name "*.db" | head -1) > $OUT/pmc_wr.txt 2>&1
grep -v "^\[" $OUT/pmc_rd.txt | grep "bh::" | head -5; grep -v "^\[" $OUT/pmc_wr.txt | grep "bh::" | head -5
rm -rf $OUT/pmc_rd $OUT/pmc_wr
