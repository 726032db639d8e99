# round-2 first GPU pass: the new -m gpu gates (C4 / C5 / full-size C2 bytes / two-rank sharded or_many), the new
# bench line (headline + config.secondary + cpu_baseline) and the realdata quick timings as the round's baseline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_distributed.py -m gpu -q -x --tb=short > gpurun_out/r2a/pytest_new.log 2>&1; tail -6 gpurun_out/r2a/pytest_new.log
timeout 900 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; cut -c1-400 gpurun_out/r2a/bench.json; tail -3 gpurun_out/r2a/bench.err
timeout 200 python scripts/quick_c3.py > gpurun_out/r2a/quick_c3.jsonl 2> gpurun_out/r2a/quick.err; cat gpurun_out/r2a/quick_c3.jsonl
timeout 60 python bench.py --gpus 2 --steps 1 > gpurun_out/r2a/bench_gpus2.out 2>&1; echo "gpus2 rc=$?"; tail -2 gpurun_out/r2a/bench_gpus2.out
