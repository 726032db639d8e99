# round 5, call c: counting-sort many-way path (count matrix, no global atomics) -- parity tests, C4 timing, timeline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5c
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "many or c4 or full_container or sharded or dense" > $O/tests.txt 2>&1; grep -E "passed|failed|error" $O/tests.txt | tail -3
for t in 8192 16384 4096; do
  echo "== RHIP_MANY_T=$t" | tee -a $O/many.txt
  RHIP_MANY_T=$t timeout 90 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-200 | tee -a $O/many.txt
done
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python scripts/prof_c4.py 100000 > $O/prof_c4.log 2>&1
python scripts/trace_many.py $O/prof_c4 "c4 or_many 100000" | tee $O/timeline_c4.txt
cp $(find $O/prof_c4 -name "*kernel_stats.csv" | head -1) $O/c4_kernel_stats.csv
rm -f $(find $O/prof_c4 -name "*kernel_trace.csv")
