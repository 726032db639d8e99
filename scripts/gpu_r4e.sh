cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4e gpurun_out/prof_r2
free -g | head -2 > gpurun_out/r4e/free.txt
# stand-alone kernel durations (one stream), grouped with chunk 8 / 4 / 2 / 16 and not grouped
for cfg in g1c8 g1c4 g1c2 g1c16 g0c8; do
g=${cfg:1:1}; ch=${cfg#*c}
for spec in w_and:and:weather_sept_85 w_or:or:weather_sept_85 w_andnot:andnot:weather_sept_85; do
  name=${spec%%:*}_$cfg; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  RHIP_NO_OVERLAP=1 RHIP_GROUP_X=$g RHIP_XG_CHUNK=$ch rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  python scripts/show_trace.py $name | grep -E "==|filter|union|wave|k_ba|usmall|probe|period"
done
done 2>&1 | tee gpurun_out/r4e/standalone.txt
for ch in 8 4; do TAG="group=1 chunk=$ch" RHIP_GROUP_X=1 RHIP_XG_CHUNK=$ch MULTI=0 timeout 200 python scripts/quick_all.py weather_sept_85 census-income 2>/dev/null | tee -a gpurun_out/r4e/quick_all.txt; done
TAG="group=0" RHIP_GROUP_X=0 MULTI=0 timeout 200 python scripts/quick_all.py weather_sept_85 census-income 2>/dev/null | tee -a gpurun_out/r4e/quick_all.txt
