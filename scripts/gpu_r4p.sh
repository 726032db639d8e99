cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p
mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python scripts/bench_line.py | tee -a $O/placed_by_library.txt
done
