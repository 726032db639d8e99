cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/bench_variants.py > gpurun_out/variants.log 2>&1
python scripts/bench_variants.py --torch >> gpurun_out/variants.log 2>&1
cat gpurun_out/variants.log
rocm-smi --showclocks --showpower 2>/dev/null | head -30
