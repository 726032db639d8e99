cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_poolops.py -m gpu -q --tb=short -k "value_lists" > gpurun_out/pytest_values.log 2>&1; tail -3 gpurun_out/pytest_values.log
timeout 200 python scripts/bench_values.py 2> gpurun_out/values.err | tee gpurun_out/values.jsonl; tail -2 gpurun_out/values.err
