cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5p
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_join.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for h in 4 2 10; do RHIP_ARENA_HOLD=$h timeout 120 python scripts/fresh_pool.py or 2>&1 | tail -1 | cut -c1-300; done
RHIP_ARENA_TRIES=0 timeout 120 python scripts/fresh_pool.py and or 2>&1 | tail -2 | cut -c1-200
LIST=1 MULTI=0 timeout 120 python scripts/quick_all.py census1881 c5 wikileaks-noquotes 2>/dev/null | tee $O/quick.txt
