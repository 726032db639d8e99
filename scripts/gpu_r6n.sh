cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6n
mkdir -p $O
S=$(date +%s); timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary > $O/bench.json 2> $O/bench.err
echo "bench wall s: $(( $(date +%s) - S ))"
python - $O/bench.json <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(round(d["value"]), round(r["frac"], 4), "traffic", r["traffic"], "ratio", round(r["traffic"] / (r["pairs_per_launch"] * 24576), 4))
print(r["traffic_source"][:700])
print(d["config"]["c2_fresh_result_pool_ms"], d["config"]["result_arena_placement"]["probe_GBps_of_each_candidate"])
P
