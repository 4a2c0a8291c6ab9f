#!/bin/bash
# the round's closing run on one GPU box: full GPU test suite, smoke, the default bench line, the bench with the driver's flags, then the
# rocprofv3 evidence for every configuration of the line (tools/r6_profiles.sh)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
TAG=${1:-r06}
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -n 4 gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; tail -n1 gpurun_out/${TAG}_bench_default.json | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_flags.json 2> gpurun_out/${TAG}_bench_driver_flags.err; tail -n1 gpurun_out/${TAG}_bench_driver_flags.json | cut -c1-300
timeout 1800 bash tools/r6_profiles.sh $TAG > gpurun_out/${TAG}_profiles.log 2>&1; grep -c "" gpurun_out/${TAG}_profiles.log
