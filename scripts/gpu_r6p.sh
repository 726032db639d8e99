cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6p
mkdir -p $O
for i in 1 2 3; do
  for v in "" oldprobe; do
    TAG="[$v]" LIST=1 MULTI=0 RHIP_LIB_VARIANT=$v timeout 200 python scripts/quick_all.py weather_sept_85 census1881 c5 2>/dev/null | tee -a $O/ab.txt
  done
done
