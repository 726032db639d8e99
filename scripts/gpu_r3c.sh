# stand-alone kernel durations (no fork, one stream): weather or / and / andnot; class stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3c
mkdir -p $O
for spec in w_or:or:weather_sept_85 w_and:and:weather_sept_85 w_andnot:andnot:weather_sept_85; do
  name=${spec%%:*}; rest=${spec#*:}; op=${rest%%:*}; ds=${rest#*:}
  RHIP_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o p -- python scripts/prof_weather.py $op $ds > $O/prof_$name.log 2>&1
  grep "min ms" $O/prof_$name.log | cut -c1-80
  python - "$O/prof_$name" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*kernel_stats.csv')
for r in list(csv.DictReader(open(f[0])))[:10] if f else []:
    print('   ', r['Name'][:50], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
P
  rm -f $O/prof_$name/*kernel_trace.csv
done
python - <<'P'
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, croaring_amd, json
from util import load_bundle, all_pairs, OPS
eng = croaring_amd.Engine(0)
pool = eng.pool_from_serialized(load_bundle("weather_sept_85"))
lhs, rhs = all_pairs(len(pool))
eng.set_class_stats(True)
for op in OPS:
    eng.pairwise(op, pool, lhs, pool, rhs)
    print(op, json.dumps({k: v for k, v in eng.last_class_stats().items() if v["items"]}))
P
