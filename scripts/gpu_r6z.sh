cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6z
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q --tb=short -x -k "placement or c2_full" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests -m gpu -q -x -k "c2_full_size_all_cardinalities or full_size_properties" > /dev/null 2>&1
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-live-traffic 2> $O/err_$i.txt | python scripts/bench_line.py | head -2 | tr "\n" " " | cut -c1-175; echo
done | tee $O/after_tests2.txt
