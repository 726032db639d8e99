cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s
mkdir -p $O gpurun_out/prof_r2
for i in 1 2; do TAG="gate" LIST=1 MULTI=1 timeout 200 python scripts/quick_all.py weather_sept_85 census-income c5 2>/dev/null | tee -a $O/quick_all.txt; done
TAG="events" RHIP_SPIN_JOIN=0 LIST=1 MULTI=1 timeout 200 python scripts/quick_all.py weather_sept_85 2>/dev/null | tee -a $O/quick_all.txt
for spec in w_or:or:weather_sept_85; do
  name=${spec%%:*}_ov; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  LIST=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  python scripts/show_trace.py $name
done 2>&1 | tee $O/timelines.txt
