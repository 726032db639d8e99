# realdata timings (prepared pair lists) of library variants built beside the product, alternating; a variant may carry
# an environment setting after '+':   gpurun -- 'bash scripts/gpu_var_realdata.sh "" i5 +RHIP_USMALL_GP8=0'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
for cfg in "$@"; do
  v=${cfg%%+*}; e=""; case "$cfg" in *+*) e=${cfg#*+};; esac
  echo "== variant '${v:-product}' $e (pass $rep)"
  env RHIP_LIB_VARIANT=$v $e LIST=1 MULTI=0 timeout 200 python scripts/quick_all.py ${DATASETS:-weather_sept_85 census-income c5} 2>/dev/null | tail -8
done; done
