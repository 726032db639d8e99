cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest14.log 2>&1; grep -E "passed|failed|Error" gpurun_out/pytest14.log | tail -3
python scripts/bench_realdata.py census1881 weather_sept_85 wikileaks-noquotes c5 > gpurun_out/realdata6.jsonl 2> gpurun_out/realdata6.err; python - <<'PY'
import json
for l in open('gpurun_out/realdata6.jsonl'):
    d=json.loads(l)
    if 'gpu_ops_per_s' in d: print(f"{d['dataset'][:28]:28s} {d['op']:16s} {d['gpu_ops_per_s']/1e6:8.2f} Mops/s  {d['gpu_ms_batch']:7.3f} ms")
PY
