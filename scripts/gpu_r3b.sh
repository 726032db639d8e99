# round 3, pass B: k_ba (bitset op array from registers), recycled many-way results -- parity, realdata timings,
# per-kernel stats of weather or / and / andnot, C4 / C5-union timings, SQ counters of k_many_l1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3b
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 300 python scripts/quick_c3.py > $O/quick_c3.jsonl 2> $O/quick.err; cat $O/quick_c3.jsonl
for spec in w_or:or:weather_sept_85 w_and:and:weather_sept_85 w_andnot:andnot:weather_sept_85; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o p -- python scripts/prof_weather.py $op $ds > $O/prof_$name.log 2>&1
  grep "min ms" $O/prof_$name.log
  python - "$O/prof_$name" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*kernel_stats.csv')
for r in list(csv.DictReader(open(f[0])))[:9] if f else []:
    print('   ', r['Name'][:50], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
P
done
timeout 120 python scripts/prof_c4.py 100000 2>&1 | tail -1
timeout 200 python - <<'P'
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, croaring_amd
from util import c5_inputs, load_bundle
eng = croaring_amd.Engine(0)
for tag, pool in (("c5_union", eng.pool_from_serialized64(c5_inputs())), ("census1881_or_many", eng.pool_from_serialized(load_bundle("census1881"))),
                  ("weather_or_many", eng.pool_from_serialized(load_bundle("weather_sept_85")))):
    ts = []
    r = None
    for _ in range(12):
        t = time.perf_counter(); r = eng.or_many(pool); ts.append(time.perf_counter() - t)
    print(tag, "min ms", round(min(ts) * 1e3, 4), "median", round(sorted(ts)[6] * 1e3, 4))
P
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_c4 -o w -- python scripts/prof_c4.py 100000 > $O/pmc_c4.log 2>&1
python - <<'P'
import collections, csv, glob
fs = glob.glob('gpurun_out/r3b/pmc_c4/*counter_collection.csv')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])) if fs else []:
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if not k.startswith("k_many") or "SQ_WAVE_CYCLES" not in c: continue
    n = len(c["SQ_WAVE_CYCLES"]); wc = sum(c["SQ_WAVE_CYCLES"]); f = lambda x: sum(c.get(x, [0]))
    print(k, n, "active %.0f%% wait_any %.0f%% wait_inst %.0f%% VALU/launch %.3g LDS/launch %.3g conflict/active %.0f%%" % (
        100 * f('SQ_ACTIVE_INST_ANY') / wc, 100 * f('SQ_WAIT_ANY') / wc, 100 * f('SQ_WAIT_INST_ANY') / wc, f('SQ_INSTS_VALU') / n, f('SQ_INSTS_LDS') / n,
        100 * f('SQ_LDS_BANK_CONFLICT') / max(1, f('SQ_LDS_IDX_ACTIVE'))))
P
rm -f $O/pmc_c4/*kernel_trace.csv
