cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r6t/pytest.log 2>&1; tail -3 gpurun_out/r6t/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
