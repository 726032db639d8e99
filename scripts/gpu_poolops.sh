cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_poolops.py -m gpu -q --tb=short > gpurun_out/pytest_poolops.log 2>&1; tail -25 gpurun_out/pytest_poolops.log
timeout 300 python scripts/bench_poolops.py > gpurun_out/poolops.jsonl 2> gpurun_out/poolops.err; cat gpurun_out/poolops.jsonl; tail -5 gpurun_out/poolops.err
