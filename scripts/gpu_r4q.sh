cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4q
mkdir -p $O gpurun_out/prof_r2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compat.py tests/test_gpu_dropin_harness.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
for sj in 1 0 1 0; do TAG="spin_join=$sj" RHIP_SPIN_JOIN=$sj LIST=1 MULTI=0 timeout 200 python scripts/quick_all.py weather_sept_85 census-income c5 2>/dev/null | tee -a $O/quick_all.txt; done
for spec in w_and:and:weather_sept_85 w_or:or:weather_sept_85; do
  name=${spec%%:*}_ov; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  LIST=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  python scripts/show_trace.py $name
done 2>&1 | tee $O/timelines.txt
