# kernel-trace profiles (rocprofv3 --kernel-trace --stats): C4 or_many, C3 weather and/or, census1881 and
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2
run() { # name, script args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/$name -o p -- python "$@" > gpurun_out/prof_r2/$name.log 2>&1
  tail -1 gpurun_out/prof_r2/$name.log
  f=$(find gpurun_out/prof_r2/$name -name "*kernel_stats.csv" | head -1)
  cut -d, -f1-7 $f | head -${HEADN:-16}
}
run c4 scripts/prof_c4.py 100000
run w_and scripts/prof_weather.py and
run w_or scripts/prof_weather.py or
HEADN=30 run c1_and scripts/prof_weather.py and census1881
