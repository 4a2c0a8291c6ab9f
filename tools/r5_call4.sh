#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q > gpurun_out/r5d_pytest.log 2>&1; tail -n 15 gpurun_out/r5d_pytest.log
timeout 300 python tools/decode_bench.py 2048 1667 1024 > gpurun_out/r5d_decode.txt 2>&1; tail -2 gpurun_out/r5d_decode.txt
timeout 300 python tools/decode_bench.py 256 2000 4096 > gpurun_out/r5d_decode_sup.txt 2>&1; tail -2 gpurun_out/r5d_decode_sup.txt
timeout 300 python tools/decode_bench.py 2048 1667 256 > gpurun_out/r5d_decode_fast.txt 2>&1; tail -2 gpurun_out/r5d_decode_fast.txt
timeout 300 python tools/decode_bench.py 512 1667 1024 > gpurun_out/r5d_decode_512.txt 2>&1; tail -2 gpurun_out/r5d_decode_512.txt
