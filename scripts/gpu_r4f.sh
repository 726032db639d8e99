cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4f gpurun_out/prof_r2
timeout 900 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | tail -5 | tee gpurun_out/r4f/pytest.txt
for g in 0 1; do TAG="dpp group=$g" RHIP_GROUP_X=$g MULTI=1 timeout 200 python scripts/quick_all.py 2>/dev/null | tee -a gpurun_out/r4f/quick_all.txt; done
for cfg in g1c8 g0c8; do
g=${cfg:1:1}; ch=${cfg#*c}
for spec in w_and:and:weather_sept_85 w_or:or:weather_sept_85 w_andnot:andnot:weather_sept_85; do
  name=${spec%%:*}_$cfg; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  RHIP_NO_OVERLAP=1 RHIP_GROUP_X=$g RHIP_XG_CHUNK=$ch rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  python scripts/show_trace.py $name
done
done 2>&1 | tee gpurun_out/r4f/standalone.txt
