# full -m gpu suite + the driver's bench command + realdata quick timings, one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/full/pytest.log 2>&1; tail -5 gpurun_out/full/pytest.log | head -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; cut -c1-300 gpurun_out/full/bench.json; tail -2 gpurun_out/full/bench.err
timeout 200 python scripts/quick_c3.py > gpurun_out/full/quick_c3.jsonl 2> gpurun_out/full/quick.err; cat gpurun_out/full/quick_c3.jsonl
