cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6l
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python scripts/bench_line.py $O/bench.json 2>/dev/null | head -40 || head -c 600 $O/bench.json
timeout 300 scripts/bin/vmm_gather > $O/vmm_gather.txt 2>&1; cat $O/vmm_gather.txt
