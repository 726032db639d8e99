cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6r
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r6r/pytest.log 2>&1; tail -3 gpurun_out/r6r/pytest.log
bash scripts/gpu_final_r6.sh > gpurun_out/r6r/final.log 2>&1; grep -c . gpurun_out/r6r/final.log
