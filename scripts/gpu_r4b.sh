cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4b
W=malloc,contig,vmm_a2m,vmm_a1g,vmm_a32g
timeout 300 scripts/bin/arena_place 5 $W > gpurun_out/r4b/place_plain.txt 2>&1
POOL_WAY=contig timeout 300 scripts/bin/arena_place 4 malloc,contig,vmm_a1g > gpurun_out/r4b/place_poolcontig.txt 2>&1
cd /tmp
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" "GRBM_UTCL2_BUSY GRBM_EA_BUSY GRBM_GUI_ACTIVE TCC_TOO_MANY_EA_WRREQS_STALL_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r4b/pmc_$tag
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4b/pmc_$tag -o p -- $GRAFT_REPO_ROOT/scripts/bin/arena_place 3 malloc,contig,vmm_a1g > $GRAFT_REPO_ROOT/gpurun_out/r4b/pmc_$tag.txt 2>&1
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/r4b/place_plain.txt gpurun_out/r4b/place_poolcontig.txt
find gpurun_out/r4b -name "*.csv" | head
