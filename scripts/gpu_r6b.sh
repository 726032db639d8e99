# round 6, step B: the interval-dominated fork (RHIP_FORK_RUNS_MIN) A/B on C5 / wikileaks / census-income + timelines with the plan cache
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6b
mkdir -p $O
for v in 0 524288 65536; do
  RHIP_FORK_RUNS_MIN=$v TAG="fork_runs_min=$v" LIST=1 MULTI=1 timeout 200 python scripts/quick_all.py c5 wikileaks-noquotes census-income 2>/dev/null | tee -a $O/quick.txt | cut -c1-200
done
timeout 300 python -m pytest tests -m gpu -q --tb=short -x -k "c5 or 64bit or realdata_all_pairs or synth_every or randomized" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for spec in c5_and:and:c5 c5_or:or:c5 w_or:or:weather_sept_85 w_and:and:weather_sept_85; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  LIST=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o p -- python scripts/prof_weather.py $op $ds > $O/prof_$name.log 2>&1
  grep "min ms" $O/prof_$name.log | cut -c1-120
  python scripts/show_trace.py $O/prof_$name 2>/dev/null | tail -22
done
rm -f $(find $O -name "*kernel_trace.csv")
echo done
