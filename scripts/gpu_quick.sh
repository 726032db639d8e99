cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "synth_every or randomized or edge or union_boundaries or extremes" > gpurun_out/pytest_quick.log 2>&1; tail -2 gpurun_out/pytest_quick.log
timeout 200 python scripts/quick_c3.py "$@" 2> gpurun_out/quick.err | tee gpurun_out/quick_c3.jsonl
