cd $GRAFT_REPO_ROOT
timeout 38 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x --tb=short -k "arena_placement_forced or full_size" 2>&1 | grep -E "passed|failed|rror" | tail -3
