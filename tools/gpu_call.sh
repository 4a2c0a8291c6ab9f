#!/bin/bash
# One parameterised GPU-box script (replaces the per-session gpu_call*.sh files): runs the named steps in order, every output under gpurun_out/.
#   usage (on the box, via gpurun): bash tools/gpu_call.sh <tag> step [step ...]
#   steps: stats2 (section cycles of the paired recurrent kernel)  tests (pytest -m gpu)
#          smoke  bench (default bench.py)  prof (tools/prof_round.sh <tag>)  q8tests  e2e
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
TAG=$1; shift
for step in "$@"; do
  case $step in
    stats2) timeout 300 python tools/lstm_stats2.py 1024 > gpurun_out/${TAG}_stats2.log 2>&1; tail -n 12 gpurun_out/${TAG}_stats2.log ;;
    tests)  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -n 5 gpurun_out/${TAG}_pytest.log ;;
    lstmtests) timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/${TAG}_pytest_lstm.log 2>&1; tail -n 5 gpurun_out/${TAG}_pytest_lstm.log ;;
    smoke)  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench)  timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -n1 gpurun_out/${TAG}_bench.json | cut -c1-3000 ;;
    benchq) timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_benchq.json 2> gpurun_out/${TAG}_benchq.err; tail -n1 gpurun_out/${TAG}_benchq.json | cut -c1-1500 ;;
    prof)   PS_ARGS="--batch 1024" timeout 900 bash tools/prof_round.sh $TAG > gpurun_out/${TAG}_prof.log 2>&1; sed -n 1,14p gpurun_out/${TAG}_prof.log | cut -c1-180 ;;
    sq)     timeout 900 bash tools/prof_sq.sh $TAG > gpurun_out/${TAG}_sq.log 2>&1; tail -n 40 gpurun_out/${TAG}_sq.log | cut -c1-160 ;;
    e2e)    timeout 600 python tools/e2e_basecall.py --reps 1 > gpurun_out/${TAG}_e2e.log 2>&1; tail -n 3 gpurun_out/${TAG}_e2e.log ;;
    e2e2)   timeout 600 taskset -c 0-1 python tools/e2e_basecall.py --reps 1 > gpurun_out/${TAG}_e2e_2cores.log 2>&1; tail -n 3 gpurun_out/${TAG}_e2e_2cores.log ;;
    cli)    timeout 900 python -m pytest tests/test_gpu_basecall.py -m gpu -x -q > gpurun_out/${TAG}_pytest_cli.log 2>&1; tail -n 4 gpurun_out/${TAG}_pytest_cli.log ;;
    paired) timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "paired or 1024" > gpurun_out/${TAG}_pytest_paired.log 2>&1; tail -n 4 gpurun_out/${TAG}_pytest_paired.log ;;
    *) echo "unknown step $step" ;;
  esac
done
