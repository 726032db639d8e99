cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest5.log 2>&1; tail -8 gpurun_out/pytest5.log
python scripts/bench_realdata.py census1881 weather_sept_85 > gpurun_out/realdata2.jsonl 2> gpurun_out/realdata2.err; cut -c1-330 gpurun_out/realdata2.jsonl; tail -3 gpurun_out/realdata2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_weather -o w -- python $GRAFT_REPO_ROOT/scripts/bench_realdata.py weather_sept_85 > $GRAFT_REPO_ROOT/gpurun_out/prof_weather.log 2>&1
