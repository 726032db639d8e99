cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4h
mkdir -p $O gpurun_out/prof_r2
timeout 1200 python -m pytest tests -m gpu -q -x --tb=short > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
# ad-hoc pair lists / prepared lists / prepared + plain result stores
TAG="adhoc" MULTI=1 timeout 200 python scripts/quick_all.py 2>/dev/null | tee -a $O/quick_all.txt
TAG="list" LIST=1 MULTI=1 timeout 200 python scripts/quick_all.py 2>/dev/null | tee -a $O/quick_all.txt
TAG="list nt0" LIST=1 MULTI=0 RHIP_LIB_VARIANT=nt0 timeout 200 python scripts/quick_all.py 2>/dev/null | tee -a $O/quick_all.txt
TAG="list copy4" LIST=1 MULTI=0 RHIP_COPY_WIDE=0 timeout 200 python scripts/quick_all.py wikileaks-noquotes c5 2>/dev/null | tee -a $O/quick_all.txt
# host phase clock without the profiler
for spec in and:weather_sept_85 or:weather_sept_85 and:census1881 or:c5; do
  op=${spec%%:*}; ds=${spec#*:}
  for l in 0 1; do echo "LIST=$l" | tee -a $O/hostclk.txt; LIST=$l python scripts/prof_weather.py $op $ds 2>&1 | grep "min ms" | cut -c1-250 | tee -a $O/hostclk.txt; done
done
# stand-alone kernel durations, grouped (two-ahead prefetch, non-temporal result stores)
for spec in w_and:and:weather_sept_85 w_or:or:weather_sept_85 w_andnot:andnot:weather_sept_85 c5_or:or:c5; do
  name=${spec%%:*}_sa; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  LIST=1 RHIP_NO_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  python scripts/show_trace.py $name
done 2>&1 | tee $O/standalone.txt
for spec in w_and:and:weather_sept_85 w_or:or:weather_sept_85 c5_or:or:c5; do
  name=${spec%%:*}_ov; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  LIST=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  python scripts/show_trace.py $name
done 2>&1 | tee $O/timelines.txt
timeout 200 scripts/bin/arena_place2 malloc,malloc,malloc,contig,malloc > $O/place2.txt 2>&1; cat $O/place2.txt
