cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6z
mkdir -p $O
for i in 1 2 3 4; do
  timeout 300 python -m pytest tests -m gpu -q -x -k "c2_full_size_all_cardinalities or full_size_properties" > /dev/null 2>&1
  RHIP_ARENA_DEBUG=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-live-traffic 2> $O/err_$i.txt | python scripts/bench_line.py | head -2 | tr "\n" " " | cut -c1-160; echo
  grep "rhip place_arena_chunks" $O/err_$i.txt | cut -c1-330
done | tee $O/after_tests.txt
