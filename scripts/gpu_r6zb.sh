cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6zb
mkdir -p $O
for i in $(seq 1 16); do
  if [ $((i % 2)) -eq 0 ]; then timeout 300 python -m pytest tests -m gpu -q -x -k "c2_full_size_all_cardinalities or full_size_properties" > /dev/null 2>&1; tag="behind a big-memory test process"; else tag="fresh"; fi
  echo "$(timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-live-traffic 2>/dev/null | python scripts/bench_line.py | head -2 | tr '\n' ' ' | cut -c1-130) ($tag)"
done | tee $O/soak.txt
