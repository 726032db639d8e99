cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/full/pytest.log 2>&1; tail -5 gpurun_out/full/pytest.log | head -3
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
