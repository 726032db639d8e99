# round-end measurement pass: everything profiles/ cites, from ONE box.  Usage: gpurun -- 'bash scripts/gpu_final.sh [--with-tests]'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
if [ "$1" = "--with-tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/final/pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/final/pytest.log | tail -3
fi
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; cut -c1-600 gpurun_out/final/bench.json
python scripts/bench_c2_ops.py > gpurun_out/final/c2_ops.jsonl 2>/dev/null
python scripts/bench_realdata.py census1881 weather_sept_85 wikileaks-noquotes census-income c5 c4=100000 > gpurun_out/final/realdata.jsonl 2> gpurun_out/final/realdata.err
python bench.py --workload ormany --steps 5 --warmup 1 > gpurun_out/final/bench_ormany.json 2>/dev/null
python scripts/bench_poolops.py > gpurun_out/final/poolops.jsonl 2>/dev/null
python scripts/quick_c3.py > gpurun_out/final/quick_c3.jsonl 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/final/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/pmc_fetch -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/final/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/pmc_write -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/final/pmc_write.log 2>&1
echo done
