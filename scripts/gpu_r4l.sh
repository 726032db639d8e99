cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l
mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --arena-tries 0 --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python scripts/bench_line.py | tee -a $O/tries0_pow2.txt
done
RHIP_ARENA_POW2=0 timeout 300 python bench.py --arena-tries 0 --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python scripts/bench_line.py | tee -a $O/tries0_nopow2.txt
RHIP_ARENA_POW2=0 timeout 300 python bench.py --arena-tries 0 --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python scripts/bench_line.py | tee -a $O/tries0_nopow2.txt
timeout 300 python bench.py --arena-tries 12 --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python scripts/bench_line.py | tee -a $O/tries12_pow2.txt
timeout 300 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "full_size or c2" 2>&1 | grep -E "passed|failed" | tail -2
