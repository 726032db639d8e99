cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for f in "" 0 16 64; do echo "== RHIP_FORK_MIN_MB=$f"; RHIP_FORK_MIN_MB=$f TAG="fork$f" LIST=1 MULTI=0 timeout 120 python scripts/quick_all.py c5 census1881 wikileaks-noquotes census-income 2>/dev/null; done
