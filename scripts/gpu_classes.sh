cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 280 python scripts/quick_classes.py 2> gpurun_out/classes.err | tee gpurun_out/classes.jsonl
tail -3 gpurun_out/classes.err
