cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_distributed.py -m gpu -q --tb=short -k "multi_rank" > $O/pytest.log 2>&1; tail -40 $O/pytest.log | cut -c1-400
