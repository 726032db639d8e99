cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./scripts/bin/bb_microbench2 2>&1 | grep -E "gap test|order 0\] product k_bb<AND> grid=4096" > gpurun_out/microbench3.log; cat gpurun_out/microbench3.log
