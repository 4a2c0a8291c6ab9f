#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r5f_pytest.log 2>&1; tail -n 6 gpurun_out/r5f_pytest.log
timeout 900 python bench.py > gpurun_out/r5f_bench.json 2> gpurun_out/r5f_bench.err; tail -n1 gpurun_out/r5f_bench.json | python -c "
import json,sys; j=json.loads(sys.stdin.read())
print('hac', j['ms_per_step'], j['value'], j['kernel_ms_per_step'], 'h2d', j['with_h2d']['ms_per_step'])
print('per_call_1', j['per_call_1']['ms_per_step'], j['per_call_1']['kernel_ms_per_step'])
print('e2e', j['e2e'])
for k,v in j['other_configs'].items(): print(k, v.get('ms_per_step'), v.get('kernel_ms_per_step'))
print('parity', {k:v for k,v in j['parity'].items() if k!='note'})
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['kind'])
"
