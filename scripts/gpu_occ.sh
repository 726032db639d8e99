# k_many_l1: the product against a variant library built beside it (RHIP_BUILD_VARIANT=name ... python -m croaring_amd.build, or a
# copy of another checkout's libroaring_hip.so as croaring_amd/libroaring_hip_<name>.so); C4, C4 x 10, PF sweep, tests.
#   gpurun -- 'bash scripts/gpu_occ.sh prev'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/occ
mkdir -p $O
V=${1:-prev}
for rep in 1 2; do
for v in "" "$V"; do
  echo "== variant '${v:-product}'"
  RHIP_LIB_VARIANT=$v timeout 200 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-200
done; done
for pf in 2 4; do echo "== product PF=$pf"; RHIP_MANY_PF=$pf timeout 200 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-200; done
echo "== product x10"; timeout 300 python scripts/prof_c4.py 1000000 2>&1 | tail -1 | cut -c1-200
echo "== $V x10"; RHIP_LIB_VARIANT=$V timeout 300 python scripts/prof_c4.py 1000000 2>&1 | tail -1 | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python scripts/prof_c4.py 100000 > $O/prof_c4.log 2>&1
python scripts/trace_many.py $O/prof_c4 "or_many product"; rm -f $(find $O/prof_c4 -name "*kernel_trace.csv")
timeout 900 python -m pytest tests -q -m gpu -k "many or sparse or c4 or sharded" > $O/tests.txt 2>&1; grep -E "passed|failed|^E  " $O/tests.txt | tail -12 | cut -c1-300
