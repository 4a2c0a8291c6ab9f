#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 300 python tools/decode_bench.py 2048 1667 1024 beam_cpw=2 beam_cpw=3 > gpurun_out/r5b_decode_btb2.txt 2>&1; cat gpurun_out/r5b_decode_btb2.txt | tail -4
BONITO_HIP_LIB=$PWD/bonito_amd/libbonito_hip_expt.so timeout 300 python tools/decode_bench.py 2048 1667 1024 beam_cpw=2 beam_cpw=4 beam_cpw=1 > gpurun_out/r5b_decode_btb1.txt 2>&1; tail -5 gpurun_out/r5b_decode_btb1.txt
BONITO_HIP_LIB=$PWD/bonito_amd/libbonito_hip_expt.so timeout 300 python tools/decode_bench.py 512 1667 1024 beam_cpw=4 > gpurun_out/r5b_decode_btb1_512.txt 2>&1; tail -3 gpurun_out/r5b_decode_btb1_512.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q > gpurun_out/r5b_pytest.log 2>&1; tail -n 8 gpurun_out/r5b_pytest.log
