cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6q
mkdir -p $O
for i in 1 2 3 4 5 6; do timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-live-traffic 2>/dev/null | python scripts/bench_line.py | head -2 | tr "\n" " "; echo; done > $O/placement_runs.txt
for i in 1 2; do RHIP_ARENA_VMM=0 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-live-traffic 2>/dev/null | python scripts/bench_line.py | head -2 | tr "\n" " "; echo "(RHIP_ARENA_VMM=0)"; done >> $O/placement_runs.txt
cat $O/placement_runs.txt | cut -c1-600
