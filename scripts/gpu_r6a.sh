# round 6, step A: the new parity tests on the MI355X + the plan cache A/B (RHIP_PLAN_CACHE=0 / 1) on the corpus
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6a
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "many or robust or frozen or prepared or dropin_many or compat or wide_bitmap or long_run" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for pc in 0 1; do
  RHIP_PLAN_CACHE=$pc TAG="plan_cache=$pc" LIST=1 MULTI=1 timeout 200 python scripts/quick_all.py 2>/dev/null | tee -a $O/quick_all.txt | cut -c1-200
done
for spec in c5_and:and:c5 c1_and:and:census1881; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  LIST=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o p -- python scripts/prof_weather.py $op $ds > $O/prof_$name.log 2>&1
  grep "min ms" $O/prof_$name.log | cut -c1-120
  python scripts/show_trace.py $O/prof_$name 2>/dev/null | tail -25
done
rm -f $(find $O -name "*kernel_trace.csv")
echo done
