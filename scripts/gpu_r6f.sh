cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6f
mkdir -p $O
timeout 300 scripts/bin/vmm_place 64 1024 > $O/vmm_64x1g.txt 2>&1; tail -12 $O/vmm_64x1g.txt
timeout 300 scripts/bin/vmm_place 64 1024 > $O/vmm_64x1g_b.txt 2>&1
timeout 300 scripts/bin/vmm_place 128 256 > $O/vmm_128x256m.txt 2>&1; tail -12 $O/vmm_128x256m.txt
