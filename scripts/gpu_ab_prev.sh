cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "" "prev"; do
  echo "== variant '${v:-product}'"
  RHIP_LIB_VARIANT=$v timeout 200 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-120
done; done
echo "== product x10"; timeout 300 python scripts/prof_c4.py 1000000 2>&1 | tail -1 | cut -c1-120
echo "== prev x10"; RHIP_LIB_VARIANT=prev timeout 300 python scripts/prof_c4.py 1000000 2>&1 | tail -1 | cut -c1-120
