cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 300 python tools/_diag.py 2>&1 | grep -v amdgpu.ids | grep -v "elements vs 128-tile kernel: 0 " > gpurun_out/r04j_diag.log; cat gpurun_out/r04j_diag.log
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "linear" > gpurun_out/r04j_pytest_linear.log 2>&1; tail -n 12 gpurun_out/r04j_pytest_linear.log
