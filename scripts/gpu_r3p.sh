cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2
MULTI=0 timeout 200 python scripts/quick_all.py 2>/dev/null
for spec in c1_and:and:census1881 w_and:and:weather_sept_85 c5_and:and:c5; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  timeout 100 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  python scripts/show_trace.py $name | grep "k_count\|k_scan\|k_emit\|period"
done
