cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6c
timeout 300 python scripts/slab_skew_c2.py 0 27 28 29 27 28 2>&1 | grep offset | tee -a gpurun_out/r6c/skew.txt
timeout 300 python scripts/slab_skew_c2.py 28 28 12 28 2>&1 | grep offset | tee -a gpurun_out/r6c/skew.txt
