# new pipeline: parity subset + realdata timings + C2 headline (no cpu / secondary)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_64bit.py -m gpu -q -x --tb=short > gpurun_out/r2b/pytest.log 2>&1; tail -4 gpurun_out/r2b/pytest.log
timeout 200 python scripts/quick_c3.py > gpurun_out/r2b/quick_c3.jsonl 2> gpurun_out/r2b/quick.err; cat gpurun_out/r2b/quick_c3.jsonl
timeout 300 python bench.py --no-cpu --no-secondary --steps 10 > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err; cut -c1-330 gpurun_out/r2b/bench.json; python -c "
import json; d=json.loads(open('gpurun_out/r2b/bench.json').read()); print(d['roofline'])"
