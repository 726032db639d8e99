cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_poolops.py -m gpu -q --tb=short -k "bulk or deserialization or select" > gpurun_out/pytest_bulk.log 2>&1; tail -3 gpurun_out/pytest_bulk.log
