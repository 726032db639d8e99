cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest16.log 2>&1; grep -E "passed|failed|Error" gpurun_out/pytest16.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
