cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest4.log 2>&1; tail -15 gpurun_out/pytest4.log
python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bench3.json 2> gpurun_out/bench3.err; cat gpurun_out/bench3.json; tail -3 gpurun_out/bench3.err
./scripts/bin/bb_microbench2 2>&1 | grep -E "order 0\] product k_bb<AND> grid=4096" | head -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof3.log 2>&1
head -12 $GRAFT_REPO_ROOT/gpurun_out/prof3/bench_kernel_stats.csv | cut -c1-60,200-
