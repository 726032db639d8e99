cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest3.log 2>&1; tail -25 gpurun_out/pytest3.log
