cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160
timeout 200 python -m pytest tests/test_gpu_poolops.py tests/test_gpu_64bit.py tests/test_gpu_compat.py -m gpu -q --tb=short > gpurun_out/pytest_head.log 2>&1; tail -3 gpurun_out/pytest_head.log
