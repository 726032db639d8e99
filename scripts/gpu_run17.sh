cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest9.log 2>&1; tail -12 gpurun_out/pytest9.log
python scripts/bench_realdata.py census1881 wikileaks-noquotes c5 > gpurun_out/realdata5.jsonl 2> gpurun_out/realdata5.err; python - <<'PY'
import json
for l in open('gpurun_out/realdata5.jsonl'):
    d=json.loads(l)
    if 'gpu_ops_per_s' in d: print(f"{d['dataset'][:28]:28s} {d['op']:16s} {d['gpu_ops_per_s']/1e6:8.2f} Mops/s  {d['gpu_ms_batch']:7.3f} ms  {d.get('gpu_GBps',0):8.1f} GB/s  cpu1 {d.get('cpu1_ops_per_s',0)/1e3:8.1f} kops/s")
    else: print(d)
PY
tail -3 gpurun_out/realdata5.err
