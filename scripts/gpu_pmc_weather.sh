cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for op in and or; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_w_$op -o w -- python scripts/prof_weather.py $op > gpurun_out/pmc_w_$op.log 2>&1
  tail -1 gpurun_out/pmc_w_$op.log
done
ls gpurun_out/pmc_w_and
