cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4c
(rocm-smi --showmemorypartition --showcomputepartition; rocm-smi --showmeminfo vram) > gpurun_out/r4c/partition.txt 2>&1
for i in 1 2; do timeout 200 scripts/bin/arena_place 3 slab:contig >> gpurun_out/r4c/slab_contig.txt 2>&1; done
timeout 200 scripts/bin/arena_place 3 slab:malloc >> gpurun_out/r4c/slab_malloc.txt 2>&1
cat gpurun_out/r4c/partition.txt | grep -v "^$" | head -30
cat gpurun_out/r4c/slab_contig.txt gpurun_out/r4c/slab_malloc.txt
# grouped queues: parity subset + timing A/B
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "grouped or (realdata_all_pairs and weather)" 2>&1 | tail -3
for g in 1 0; do TAG="group=$g" RHIP_GROUP_X=$g MULTI=1 timeout 200 python scripts/quick_all.py 2>/dev/null | tee -a gpurun_out/r4c/quick_all.txt; done
