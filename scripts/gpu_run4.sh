set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./scripts/bin/bb_microbench2 > gpurun_out/microbench2.log 2>&1; cat gpurun_out/microbench2.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch $GRAFT_REPO_ROOT/gpurun_out/pmc_write
