# ROUND 5, first experiment (prepared in round 4, not yet run): occupancy variants of k_many_l1 / k_union_g.
# On the CPU side first (the variant libraries travel with the snapshot):
#   RHIP_BUILD_VARIANT=many5  RHIP_EXTRA_FLAGS=-DRHIP_MANY_WAVES=5  python -m croaring_amd.build
#   RHIP_BUILD_VARIANT=union5 RHIP_EXTRA_FLAGS=-DRHIP_UNION_WAVES=5 python -m croaring_amd.build
# then: gpurun --timeout 300 -- 'bash scripts/gpu_r5a.sh'      (every command under `timeout`)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
for v in "" many5; do
  echo "== k_many_l1 variant '${v:-product}'" | tee -a $O/many.txt
  RHIP_LIB_VARIANT=$v timeout 60 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-200 | tee -a $O/many.txt
done
for v in "" union5; do
  TAG="union variant '${v:-product}'" RHIP_LIB_VARIANT=$v LIST=1 MULTI=0 timeout 60 python scripts/quick_all.py weather_sept_85 2>/dev/null | tee -a $O/union.txt
done
