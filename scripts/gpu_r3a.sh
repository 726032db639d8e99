# round 3, pass A: the rewritten many-way path on the MI355X -- full -m gpu suite, C4 timings for the prefetch depths,
# kernel stats of C4, the bench line (no CPU leg)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log | head -4
for pf in 2 4 8; do RHIP_MANY_PF=$pf timeout 120 python scripts/prof_c4.py 100000 2>&1 | tail -1 | sed "s/^/pf=$pf /"; done | tee $O/c4_pf.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python scripts/prof_c4.py 100000 > $O/prof_c4.log 2>&1; tail -1 $O/prof_c4.log | cut -c1-200
python - <<'P'
import csv, glob
f = glob.glob('gpurun_out/r3a/prof_c4/*kernel_stats.csv')
if f:
    for r in list(csv.DictReader(open(f[0])))[:16]:
        print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
P
rm -f $O/prof_c4/*kernel_trace.csv
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu > $O/bench.json 2> $O/bench.err; python - <<'P'
import json
try:
    j = json.loads(open('gpurun_out/r3a/bench.json').read())
    print('value', j['value'], 'frac', j['roofline']['frac'])
    for k, v in j['config']['secondary'].items():
        if isinstance(v, dict): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ('ms_batch_median', 'ms_batch_pipelined2', 'alg_GBps', 'frac', 'ms_median', 'ms_min', 'checksum_ok', 'sharded_w1', 'sharded_w1_nccl', 'sharded_w1_error', 'cardinality_ok')})
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/r3a/bench.err').read()[-1500:])
P
