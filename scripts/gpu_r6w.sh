cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6w
mkdir -p $O
for i in 1 2 3 4; do
  for cfg in "3 1" "2 2" "3 2"; do
    set -- $cfg
    echo "runs $1 stride $2: $(RHIP_ARENA_CHUNK_RUNS=$1 RHIP_ARENA_CHUNK_STRIDE=$2 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-live-traffic 2>/dev/null | python scripts/bench_line.py | head -2 | tr '\n' ' ' | cut -c1-150)"
  done
done | tee $O/probe_precision.txt
