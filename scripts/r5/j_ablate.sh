cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5j
mkdir -p $O
for v in l17 l18 l11 l12; do
  RHIP_LIB_VARIANT=$v timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python scripts/prof_c4.py 100000 > $O/prof_$v.log 2>&1
  python scripts/trace_many.py $O/prof_$v "variant '${v:-product}'" | tee -a $O/timelines.txt
  rm -f $(find $O/prof_$v -name "*kernel_trace.csv")
done
