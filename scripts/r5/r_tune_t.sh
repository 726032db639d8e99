cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for t in 0 65536 32768; do echo "== c4x10 T=$t"; RHIP_MANY_T=$t timeout 200 python scripts/prof_c4.py 1000000 2>&1 | tail -1 | cut -c1-150; done
for t in 0 6144 4096; do echo "== c4 T=$t"; RHIP_MANY_T=$t timeout 100 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-150; done
