cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "many or shard or edge or full_size" > gpurun_out/pytest10.log 2>&1; tail -5 gpurun_out/pytest10.log
cd /tmp && export TMPDIR=/tmp
SKIP_CPU=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/scripts/bench_realdata.py c4=100000 > $GRAFT_REPO_ROOT/gpurun_out/prof_c4.log 2>&1
grep '"dataset"' $GRAFT_REPO_ROOT/gpurun_out/prof_c4.log | cut -c1-420
