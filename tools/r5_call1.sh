#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_basecall.py -m gpu -x -q > gpurun_out/r5a_pytest.log 2>&1; tail -n 5 gpurun_out/r5a_pytest.log
timeout 900 python bench.py > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err; tail -n1 gpurun_out/r5a_bench.json | cut -c1-6000
tail -5 gpurun_out/r5a_bench.err
timeout 200 python tools/gemm_bench.py 853504,256,96,0 853504,1024,384,0 > gpurun_out/r5a_gemm_fast_head.txt 2>&1; tail -8 gpurun_out/r5a_gemm_fast_head.txt
