# C4 whole-call timings of the product under an environment sweep, three passes:  gpurun -- 'bash scripts/gpu_ab_c4.sh RHIP_MANY_PF 1 2 3 4'
cd $GRAFT_REPO_ROOT
var=$1; shift
for rep in 1 2 3; do
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v timeout 200 python scripts/prof_c4.py ${N:-100000} 2>&1 | tail -1 | cut -c1-120
done; done
