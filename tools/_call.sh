cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "linear" > gpurun_out/r04a_pytest_linear.log 2>&1; tail -n 15 gpurun_out/r04a_pytest_linear.log
timeout 300 python tools/gemm_bench.py > gpurun_out/r04a_gemm_bench.log 2>&1; cat gpurun_out/r04a_gemm_bench.log
