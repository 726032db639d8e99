cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final_r4
mkdir -p $O
timeout 45 python scripts/bench_c2_ops.py > $O/c2_ops.jsonl 2> $O/c2_ops.err; echo "c2_ops rc=$?"; cut -c1-160 $O/c2_ops.jsonl; tail -2 $O/c2_ops.err | cut -c1-200
for op in and or; do
  (cd /tmp && LIST=1 timeout 40 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_w_$op -o w -- python $GRAFT_REPO_ROOT/scripts/prof_weather.py $op > $GRAFT_REPO_ROOT/$O/pmc_w_$op.log 2>&1; echo "pmc $op rc=$?")
done
rm -f $O/pmc_w_*/*kernel_trace.csv
