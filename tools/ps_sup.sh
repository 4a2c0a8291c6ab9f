cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/ps_sup -o ps -- python $R/tools/profile_step.py --model sup --steps 2 > $R/gpurun_out/ps_sup.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/ps_sup -name "*.db" | head -1) $R/gpurun_out/ps_sup.csv | head -14
rm -rf $R/gpurun_out/ps_sup
