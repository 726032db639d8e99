# round 5, call a: C4 or_many baseline -- product vs the 5-waves variant of k_many_l1, then the SQ counters of the product
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
for v in "" many5; do
  echo "== k_many_l1 variant '${v:-product}'" | tee -a $O/many.txt
  RHIP_LIB_VARIANT=$v timeout 90 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-200 | tee -a $O/many.txt
  RHIP_LIB_VARIANT=$v timeout 90 python scripts/prof_c4.py 100000 2>&1 | tail -1 | cut -c1-200 | tee -a $O/many.txt
done
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_c4 -o w -- python scripts/prof_c4.py 100000 > $O/pmc_c4.log 2>&1
tail -2 $O/pmc_c4.log | cut -c1-200
python scripts/summarize_sq_c4.py $O/pmc_c4 | tee $O/pmc_c4_sq.md
rm -f $O/pmc_c4/*/*kernel_trace.csv $O/pmc_c4/*kernel_trace.csv
