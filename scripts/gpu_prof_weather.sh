cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for op in and or; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_w_$op -o w -- python scripts/prof_weather.py $op > gpurun_out/prof_w_$op.log 2>&1
  tail -2 gpurun_out/prof_w_$op.log
  f=$(find gpurun_out/prof_w_$op -name "*kernel_stats.csv" | head -1)
  cut -d, -f1-7 $f | head -24
done
