# grid cap / fork threshold matrix on the small realdata batches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3d
mkdir -p $O
for cap in 0 1280 2560; do for fk in 144 100000; do
  TAG="cap=$cap fork_min=$fk" RHIP_GRID_CAP=$cap RHIP_FORK_MIN_MB=$fk timeout 120 python scripts/quick_all.py 2>/dev/null
done; done | tee $O/matrix.txt
