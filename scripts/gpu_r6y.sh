cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6y
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "placement or c2_full or full_size or inplace or dropin_pairwise" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2 3 4; do timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --no-live-traffic 2>/dev/null | python scripts/bench_line.py | head -2 | tr "\n" " " | cut -c1-200; echo; done
