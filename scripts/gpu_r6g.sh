cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6g
mkdir -p $O
timeout 300 scripts/bin/vmm_place2 40 96 1024 > $O/vmm2_a.txt 2>&1; cat $O/vmm2_a.txt
timeout 300 scripts/bin/vmm_place2 40 96 1024 > $O/vmm2_b.txt 2>&1; cat $O/vmm2_b.txt
