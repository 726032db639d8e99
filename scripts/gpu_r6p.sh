cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6p
mkdir -p $O
for i in 1 2; do
  for v in "" ablx; do
    RHIP_LIB_VARIANT=$v timeout 200 python scripts/ab_realdata.py weather_sept_85 or xor 2>/dev/null | tee -a $O/ab5.txt
  done
done
