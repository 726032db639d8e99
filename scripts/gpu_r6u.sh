cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6u
mkdir -p $O
timeout 300 scripts/bin/vmm_place5 16 > $O/vmm5_a.txt 2>&1; cat $O/vmm5_a.txt
timeout 300 scripts/bin/vmm_place5 16 > $O/vmm5_b.txt 2>&1; tail -5 $O/vmm5_b.txt
