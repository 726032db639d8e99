cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for dyn in 0 1; do
  if [ $dyn = 1 ]; then export RHIP_DYN=1; else unset RHIP_DYN; fi
  echo "== RHIP_DYN=$dyn"
  python scripts/bench_realdata.py weather_sept_85 wikileaks-noquotes c5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l)
    if 'gpu_ops_per_s' in d: print(' ', d['dataset'][:22], d['op'], round(d['gpu_ops_per_s']/1e6,2), 'Mops/s', round(d['gpu_ms_batch'],3), 'ms')
"
done
unset RHIP_DYN
python bench.py --steps 6 --warmup 2 --no-cpu --pairs 500 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pairs500', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
python bench.py --steps 12 --warmup 2 --no-cpu --pairs 250 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pairs250', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
