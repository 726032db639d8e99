cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_final.log 2>&1; grep -E "passed|failed|Error|skipped" gpurun_out/pytest_final.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | cut -c1-400
