cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest7.log 2>&1; tail -6 gpurun_out/pytest7.log
python scripts/bench_realdata.py census1881 weather_sept_85 > gpurun_out/realdata4.jsonl 2> gpurun_out/realdata4.err; python - <<'PY'
import json
for l in open('gpurun_out/realdata4.jsonl'):
    d=json.loads(l)
    if 'gpu_ops_per_s' in d: print(f"{d['dataset']:20s} {d['op']:16s} {d['gpu_ops_per_s']/1e6:8.2f} Mops/s  {d['gpu_ms_batch']:7.3f} ms  {d.get('gpu_GBps',0):8.1f} GB/s  cpu1 {d.get('cpu1_ops_per_s',0)/1e3:8.1f} kops/s")
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_weather3 -o w -- python $GRAFT_REPO_ROOT/scripts/bench_realdata.py weather_sept_85 > $GRAFT_REPO_ROOT/gpurun_out/prof_weather3.log 2>&1
head -8 $GRAFT_REPO_ROOT/gpurun_out/prof_weather3/w_kernel_stats.csv | cut -c1-90
