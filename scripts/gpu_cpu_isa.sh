cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 100 python scripts/cpu_isa_variants.py 1.2 2> gpurun_out/cpu_isa.err | tee gpurun_out/cpu_isa.jsonl
