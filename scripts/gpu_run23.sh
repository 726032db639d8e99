cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest11.log 2>&1; grep -E "passed|failed|Error" gpurun_out/pytest11.log | tail -5
python bench.py --workload ormany --steps 5 --warmup 1 > gpurun_out/bench_ormany.json 2> gpurun_out/bench_ormany.err; cat gpurun_out/bench_ormany.json; tail -2 gpurun_out/bench_ormany.err
