cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6x
mkdir -p $O
for i in $(seq 1 14); do
  RHIP_ARENA_DEBUG=1 timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu --no-secondary --no-live-traffic 2> $O/err_$i.txt | python scripts/bench_line.py | head -1 | cut -c1-80
  grep "rhip place_arena_chunks" $O/err_$i.txt | cut -c1-700
done | tee $O/debug.txt
