# round 5, call d: C4 timelines of the many-way path for several workgroup shares (RHIP_MANY_T)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
for t in 8192 32768; do
  RHIP_MANY_T=$t timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4_$t -o p -- python scripts/prof_c4.py 100000 > $O/prof_c4_$t.log 2>&1
  tail -1 $O/prof_c4_$t.log | cut -c1-200
  python scripts/trace_many.py $O/prof_c4_$t "c4 or_many 100000 T=$t" | tee -a $O/timeline_c4.txt
  rm -f $(find $O/prof_c4_$t -name "*kernel_trace.csv")
done
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "many or c4 or full_container or sharded or dense" > $O/tests.txt 2>&1; grep -E "passed|failed|error" $O/tests.txt | tail -3
