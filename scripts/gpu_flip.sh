cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_poolops.py -m gpu -q --tb=short -k "flip or value_lists" > gpurun_out/pytest_flip.log 2>&1; tail -3 gpurun_out/pytest_flip.log
