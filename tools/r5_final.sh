#!/bin/bash
# the round's closing run on a GPU box: full GPU test suite, smoke, the default bench line, the bench with the driver's flags, the rocprofv3
# round profile (kernel trace of the bench command + HBM PMC passes) and the decode-stage profile
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
TAG=${1:-r05}
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -n 4 gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -n1 gpurun_out/${TAG}_bench.json | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_flags.json 2> gpurun_out/${TAG}_bench_driver_flags.err; tail -n1 gpurun_out/${TAG}_bench_driver_flags.json | cut -c1-300
PS_ARGS="--batch 1024" timeout 900 bash tools/prof_round.sh $TAG > gpurun_out/${TAG}_prof.log 2>&1; sed -n 1,30p gpurun_out/${TAG}_prof.log | cut -c1-170
timeout 600 bash tools/prof_decode.sh $TAG > gpurun_out/${TAG}_prof_decode.log 2>&1; head -5 gpurun_out/${TAG}_prof_decode.log
# the N > 1 launch paths on this one-GPU box (ranks share the device, the two collectives go over gloo): bench.py's own spawner and the
# driver's launcher line
timeout 600 python bench.py --gpus 2 --steps 8 --warmup 4 > gpurun_out/${TAG}_bench_gpus2_self.json 2> gpurun_out/${TAG}_bench_gpus2_self.err; tail -n1 gpurun_out/${TAG}_bench_gpus2_self.json | cut -c1-260
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 8 --warmup 4 > gpurun_out/${TAG}_bench_gpus2_torchrun.json 2> gpurun_out/${TAG}_bench_gpus2_torchrun.err; tail -n1 gpurun_out/${TAG}_bench_gpus2_torchrun.json | cut -c1-260
