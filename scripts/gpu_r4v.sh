cd $GRAFT_REPO_ROOT
timeout 110 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_poolops.py -m gpu -q -x --tb=short -k "arena_placement_forced or distributed or sharded or poolops or inplace or flip or deser or select" 2>&1 | tail -6
