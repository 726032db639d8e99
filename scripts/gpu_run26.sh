cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "synth or many or random or edge or extremes" > gpurun_out/pytest15.log 2>&1; grep -E "passed|failed|Error" gpurun_out/pytest15.log | tail -3
python scripts/bench_realdata.py weather_sept_85 > gpurun_out/realdata7.jsonl 2> gpurun_out/realdata7.err; python - <<'PY'
import json
for l in open('gpurun_out/realdata7.jsonl'):
    d=json.loads(l)
    if 'gpu_ops_per_s' in d: print(f"{d['dataset'][:28]:28s} {d['op']:16s} {d['gpu_ops_per_s']/1e6:8.2f} Mops/s  {d['gpu_ms_batch']:7.3f} ms")
PY
SKIP_CPU=1 python scripts/bench_realdata.py c4=100000 2>/dev/null | cut -c1-330
