cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/bench_realdata.py census1881 weather_sept_85 wikileaks-noquotes census-income c4=100000 > gpurun_out/realdata1.jsonl 2> gpurun_out/realdata1.err; cat gpurun_out/realdata1.jsonl; tail -5 gpurun_out/realdata1.err
