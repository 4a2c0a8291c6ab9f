cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in fast hac random256; do
  rocprofv3 --kernel-trace -d $R/gpurun_out/dc_$c -o dc -- python $R/tools/decode_case.py $c > $R/gpurun_out/dc_$c.log 2>&1
  grep -E "rep|scores|emitted" $R/gpurun_out/dc_$c.log
  python $R/tools/rocprof_summary.py $(find $R/gpurun_out/dc_$c -name "*.db" | head -1) $R/gpurun_out/dc_$c.csv | grep -i "beam\|crf_" 
done
