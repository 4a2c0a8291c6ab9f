#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_decode.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_basecall.py -m gpu -q > gpurun_out/r5e_pytest.log 2>&1; tail -n 15 gpurun_out/r5e_pytest.log
