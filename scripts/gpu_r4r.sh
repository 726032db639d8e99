cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4r
mkdir -p $O gpurun_out/prof_r2
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "realdata_all_pairs or explicit_unit or grouped or batches_in_flight or multi or class_stats" > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
for i in 1 2; do TAG="crit-on-main" LIST=1 MULTI=1 timeout 200 python scripts/quick_all.py 2>/dev/null | tee -a $O/quick_all.txt; done
for spec in w_and:and:weather_sept_85 w_or:or:weather_sept_85; do
  name=${spec%%:*}_ov; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  LIST=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  python scripts/show_trace.py $name
done 2>&1 | tee $O/timelines.txt
