cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6za
mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do timeout 120 scripts/bin/vmm_place6 2>&1 | grep -v "^pool" ; echo; done | tee $O/vmm6.txt
