cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/bench_c2_ops.py > gpurun_out/c2_ops.jsonl 2>/dev/null; cat gpurun_out/c2_ops.jsonl
python scripts/bench_realdata.py wikileaks-noquotes c5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l)
    if 'gpu_ops_per_s' in d: print(d['dataset'][:24], d['op'], round(d['gpu_ops_per_s']/1e6,2), 'Mops/s', round(d['gpu_ms_batch'],3), 'ms')
"
