cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
SKIP_CPU=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/scripts/bench_realdata.py c4=100000 > $GRAFT_REPO_ROOT/gpurun_out/prof_c4.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_c4.log | cut -c1-400
