# kernel timeline of pipelined (two in flight) all-pairs batches: name:op:dataset ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  python scripts/prof_weather.py $op $ds pipe | grep "min ms"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds pipe > gpurun_out/prof_r2/$name.log 2>&1
  grep "min ms" gpurun_out/prof_r2/$name.log
done
