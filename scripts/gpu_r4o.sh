cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o
mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python scripts/bench_line.py | tee -a $O/placed_by_library.txt
done
RHIP_ARENA_TRIES=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python scripts/bench_line.py | tee -a $O/tries0.txt
RHIP_ARENA_TRIES=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python scripts/bench_line.py | tee -a $O/tries0.txt
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tail -2
