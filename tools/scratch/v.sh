cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_parity.py tests/test_gpu_basecall.py -q -m gpu -x 2>&1 | tail -4
for m in hac fast sup; do timeout 200 python bench.py --model $m --decoder viterbi --no-side-legs --no-cpu-baseline --no-h2d-leg 2>/dev/null | tail -1 > gpurun_out/r05k_bench_viterbi_$m.json; python -c "
import json; d=json.loads(open(\"gpurun_out/r05k_bench_viterbi_$m.json\").read()); print(\"$m\", round(d[\"ms_per_step\"],3), \"%.4g\" % d[\"value\"], d[\"kernel_ms_per_step\"])"; done
