cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5q
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_distributed.py -q -m gpu -k "c_abi" > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^E  " $O/tests.txt | head -12 | cut -c1-300
for h in 0; do timeout 120 python scripts/fresh_pool.py and or 2>&1 | tail -2 | cut -c1-300; done
