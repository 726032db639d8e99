cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4t
mkdir -p $O
TAG="final" LIST=1 MULTI=0 timeout 40 python scripts/quick_all.py weather_sept_85 census1881 2>&1 | tail -2 | tee $O/quick_all.txt
(cd /tmp && LIST=1 timeout 50 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_w_and -o w -- python $GRAFT_REPO_ROOT/scripts/prof_weather.py and > $GRAFT_REPO_ROOT/$O/pmc_w_and.log 2>&1; echo "pmc rc=$?"; grep "min ms" $GRAFT_REPO_ROOT/$O/pmc_w_and.log | cut -c1-80)
timeout 45 python scripts/bench_c2_ops.py > $O/c2_ops.jsonl 2> $O/c2_ops.err; echo "c2_ops rc=$?"; tail -3 $O/c2_ops.err | cut -c1-300; cut -c1-150 $O/c2_ops.jsonl
timeout 70 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "(realdata_all_pairs and weather) or explicit_unit_arrays or grouped" 2>&1 | grep -E "passed|failed|error" | tail -2
