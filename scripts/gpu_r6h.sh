cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6h
mkdir -p $O
timeout 400 scripts/bin/vmm_place3 > $O/vmm3_a.txt 2>&1; cat $O/vmm3_a.txt
