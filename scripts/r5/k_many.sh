cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5k
mkdir -p $O
for v in "" sc3; do
RHIP_LIB_VARIANT=$v timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4$v -o p -- python scripts/prof_c4.py 100000 > $O/prof_c4$v.log 2>&1
tail -1 $O/prof_c4$v.log | cut -c1-200
python scripts/trace_many.py $O/prof_c4$v "c4 or_many 100000 variant '$v'" | tee -a $O/timeline_c4.txt
rm -f $(find $O/prof_c4$v -name "*kernel_trace.csv")
done
for pf in 2; do echo "== PF $pf"; RHIP_MANY_PF=$pf timeout 90 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-200; done
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "many or c4 or full_container or sharded or dense" > $O/tests.txt 2>&1; grep -E "passed|failed|error" $O/tests.txt | tail -3
