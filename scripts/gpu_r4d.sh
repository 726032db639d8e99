cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4d gpurun_out/prof_r2
VARIANTS=1 timeout 300 scripts/bin/arena_place 2 slab:contig > gpurun_out/r4d/variants_contig.txt 2>&1
cat gpurun_out/r4d/variants_contig.txt
for g in 1 0; do
for spec in w_and:and:weather_sept_85 w_or:or:weather_sept_85 w_andnot:andnot:weather_sept_85 ci_andnot:andnot:census-income; do
  name=${spec%%:*}_g$g; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rm -rf gpurun_out/prof_r2/$name
  RHIP_GROUP_X=$g rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  tail -1 gpurun_out/prof_r2/$name.log
  python scripts/show_trace.py $name
done
done 2>&1 | tee gpurun_out/r4d/timelines.txt
for g in 1 0; do TAG="group=$g" RHIP_GROUP_X=$g MULTI=0 timeout 200 python scripts/quick_all.py 2>/dev/null | tee -a gpurun_out/r4d/quick_all.txt; done
