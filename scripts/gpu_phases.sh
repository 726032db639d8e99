cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 280 python scripts/quick_phases.py 2> gpurun_out/phases.err | tee gpurun_out/phases.jsonl
tail -3 gpurun_out/phases.err
