# k_many_l1 at five waves per SIMD (product) against a four-wave variant build (RHIP_BUILD_VARIANT=w4 RHIP_EXTRA_FLAGS=-DRHIP_MANY_WAVES=4
# python -m croaring_amd.build), each with the piece count that fills it; C4, C4 x 10, kernel timeline, many-way tests.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/occ
mkdir -p $O
for rep in 1 2; do
for cfg in ":1280" "w4:1024"; do
  v=${cfg%%:*}; s=${cfg##*:}
  echo "== variant '${v:-product}' slots $s"
  RHIP_LIB_VARIANT=$v RHIP_MANY_SLOTS=$s timeout 200 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-200
done; done
echo "== product PF=4"; RHIP_MANY_PF=4 timeout 200 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-200
echo "== product x10"; timeout 300 python scripts/prof_c4.py 1000000 2>&1 | tail -1 | cut -c1-200
echo "== w4 x10"; RHIP_LIB_VARIANT=w4 RHIP_MANY_SLOTS=1024 timeout 300 python scripts/prof_c4.py 1000000 2>&1 | tail -1 | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python scripts/prof_c4.py 100000 > $O/prof_c4.log 2>&1
python scripts/trace_many.py $O/prof_c4 "or_many product"; rm -f $(find $O/prof_c4 -name "*kernel_trace.csv")
timeout 900 python -m pytest tests -q -m gpu -k "many or sparse or c4 or sharded" > $O/tests.txt 2>&1; grep -E "passed|failed|^E  " $O/tests.txt | tail -12 | cut -c1-300
