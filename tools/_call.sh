cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 300 python tools/_diag.py 2>&1 | grep -v amdgpu.ids | grep -v "elements vs 128-tile kernel: 0 " > gpurun_out/r04p_diag.log; head -5 gpurun_out/r04p_diag.log; wc -l gpurun_out/r04p_diag.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04p_pytest.log 2>&1; tail -n 6 gpurun_out/r04p_pytest.log
timeout 300 python tools/gemm_bench.py > gpurun_out/r04p_gemm_bench.log 2>&1; cat gpurun_out/r04p_gemm_bench.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04p_bench.json 2> gpurun_out/r04p_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04p_bench.json').read().strip().split('\n')[-1])
print("hac", d['ms_per_step'], d['kernel_ms_per_step'])
for k,v in d['other_configs'].items(): print(k, v['ms_per_step'], v['kernel_ms_per_step'])
PY
