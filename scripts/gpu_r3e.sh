# kernel timelines of one batch: census1881 and / or, c5 and, wikileaks and, weather and / or
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2
for spec in c1_and:and:census1881 c1_or:or:census1881 c5_and:and:c5 wk_and:and:wikileaks-noquotes w_and:and:weather_sept_85 w_or:or:weather_sept_85; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  grep "min ms" gpurun_out/prof_r2/$name.log | cut -c1-220
done
python scripts/show_trace.py c1_and c1_or c5_and wk_and w_and w_or | tee gpurun_out/r3e_timelines.txt
