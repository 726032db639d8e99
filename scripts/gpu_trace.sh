# kernel timelines only; env passes through (e.g. RHIP_NO_OVERLAP=1); args: name:op:dataset ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/$name -o p -- python scripts/prof_weather.py $op $ds > gpurun_out/prof_r2/$name.log 2>&1
  grep "min ms" gpurun_out/prof_r2/$name.log
done
