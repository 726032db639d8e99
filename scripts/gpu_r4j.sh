cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4j
mkdir -p $O
SIZE_MODE=bench timeout 200 scripts/bin/arena_place 10 malloc > $O/place_benchsize.txt 2>&1; grep trial $O/place_benchsize.txt
SIZE_MODE=exact timeout 200 scripts/bin/arena_place 10 malloc > $O/place_exact.txt 2>&1; grep trial $O/place_exact.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-x10 > $O/bench.json 2> $O/bench.err; python - <<'P'
import json
d = json.load(open("gpurun_out/r4j/bench.json"))
print(d["config"]["secondary_summary"]["rows"]["c4_shard_stages"], d["config"]["result_arena_startup"]["k_bb_ms_of_each_try"])
P
