cd $GRAFT_REPO_ROOT
timeout 80 python -m pytest tests/test_gpu_configs.py -m gpu -q -x --tb=short 2>&1 | grep -E "passed|failed|error|Error" | tail -4
