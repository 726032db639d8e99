cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m
mkdir -p $O
SLAB_GIB=64 timeout 200 scripts/bin/arena_place 2 slab:malloc 2>&1 | tee $O/slab64_malloc.txt | cut -c1-400
SLAB_GIB=64 timeout 200 scripts/bin/arena_place 2 slab:contig 2>&1 | tee $O/slab64_contig.txt | cut -c1-400
SLAB_GIB=128 timeout 200 scripts/bin/arena_place 1 slab:malloc 2>&1 | tee $O/slab128_malloc.txt | cut -c1-600
