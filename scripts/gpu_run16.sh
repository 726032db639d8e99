cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest8.log 2>&1; tail -6 gpurun_out/pytest8.log
python scripts/bench_realdata.py c5 > gpurun_out/c5.jsonl 2> gpurun_out/c5.err; cat gpurun_out/c5.jsonl; tail -3 gpurun_out/c5.err
