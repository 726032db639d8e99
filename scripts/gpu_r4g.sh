cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4g
mkdir -p $O gpurun_out/prof_r2
timeout 900 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | tail -5 | tee $O/pytest.txt
for g in 0 1; do TAG="group=$g" RHIP_GROUP_X=$g MULTI=1 timeout 200 python scripts/quick_all.py 2>/dev/null | tee -a $O/quick_all.txt; done
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python scripts/bench_line.py < $O/bench.json | cut -c1-600
# stand-alone kernel durations on weather, grouped and not
for cfg in g1c8 g0c8; do
g=${cfg:1:1}; ch=${cfg#*c}
for spec in w_and:and:weather_sept_85 w_or:or:weather_sept_85 w_andnot:andnot:weather_sept_85; do
  name=${spec%%:*}_$cfg; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  RHIP_NO_OVERLAP=1 RHIP_GROUP_X=$g RHIP_XG_CHUNK=$ch rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  python scripts/show_trace.py $name
done
done 2>&1 | tee $O/standalone.txt
# overlapped timelines (the product schedule), grouped
for spec in w_and:and:weather_sept_85 w_or:or:weather_sept_85 c1_and:and:census1881 c5_or:or:c5; do
  name=${spec%%:*}_ov; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  python scripts/show_trace.py $name
done 2>&1 | tee $O/timelines.txt
# result-arena placement
timeout 200 scripts/bin/arena_place 4 malloc,contig,vmm_a1g > $O/place_plain.txt 2>&1
VARIANTS=1 timeout 200 scripts/bin/arena_place 2 slab:contig > $O/slab_contig.txt 2>&1
timeout 100 scripts/bin/arena_place 2 slab:malloc > $O/slab_malloc.txt 2>&1
cat $O/place_plain.txt $O/slab_contig.txt $O/slab_malloc.txt
