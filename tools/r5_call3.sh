#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r5c_pytest.log 2>&1; tail -n 6 gpurun_out/r5c_pytest.log
timeout 300 python tools/decode_bench.py 2048 1667 1024 beam_cpw=1 > gpurun_out/r5c_decode.txt 2>&1; tail -3 gpurun_out/r5c_decode.txt
timeout 300 python tools/decode_bench.py 1024 1667 1024 beam_cpw=4 > gpurun_out/r5c_decode1024.txt 2>&1; tail -3 gpurun_out/r5c_decode1024.txt
timeout 300 python tools/decode_bench.py 256 2000 4096 > gpurun_out/r5c_decode_sup.txt 2>&1; tail -2 gpurun_out/r5c_decode_sup.txt
timeout 300 python tools/decode_bench.py 1024 2000 4096 > gpurun_out/r5c_decode_sup1024.txt 2>&1; tail -2 gpurun_out/r5c_decode_sup1024.txt
timeout 300 python tools/decode_bench.py 2048 1667 256 > gpurun_out/r5c_decode_fast.txt 2>&1; tail -2 gpurun_out/r5c_decode_fast.txt
timeout 600 python bench.py --no-cpu-baseline --no-side-legs > gpurun_out/r5c_bench.json 2> gpurun_out/r5c_bench.err; tail -n1 gpurun_out/r5c_bench.json | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline --no-side-legs --model sup --steps 12 --warmup 3 > gpurun_out/r5c_bench_sup.json 2> gpurun_out/r5c_bench_sup.err; tail -n1 gpurun_out/r5c_bench_sup.json | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['kernel_ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-side-legs --model sup --steps 12 --warmup 4 --per-call 4 > gpurun_out/r5c_bench_sup4.json 2> gpurun_out/r5c_bench_sup4.err; tail -n1 gpurun_out/r5c_bench_sup4.json | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['kernel_ms_per_step'])"; tail -3 gpurun_out/r5c_bench_sup4.err
