# quick: realdata timings + timelines of weather and / or / andnot after a scheduling change
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2
TAG=now timeout 200 python scripts/quick_all.py 2>/dev/null
for spec in ${SPECS:-w_and:and:weather_sept_85 w_or:or:weather_sept_85 w_andnot:andnot:weather_sept_85}; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  python scripts/show_trace.py $name
done
