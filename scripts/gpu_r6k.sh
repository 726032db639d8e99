cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6j
mkdir -p $O
for v in 462 206; do
  timeout 120 scripts/bin/vmm_place4 $v 12 > $O/v$v.txt 2>&1
  echo "variant $v: $(grep -c '^ok' $O/v$v.txt) $(grep -o 'Memory access fault.*' $O/v$v.txt | cut -c1-120)"; grep -o '\[[a-z]* *dev_bad[^]]*\]' $O/v$v.txt | sort | uniq -c; grep re-reserved $O/v$v.txt
done
timeout 600 python -m pytest tests -m gpu -q --tb=short -x -k "placement or c2_full" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2 3 4; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary > $O/bench_vmm_$i.json 2> $O/bench_vmm_$i.err
  python - $O/bench_vmm_$i.json <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print("vmm", round(d["value"]), round(d["roofline"]["frac"], 4), d["config"]["c2_fresh_result_pool_ms"]["and"], d["config"]["c2_fresh_result_pool_ms"]["or"], d["config"]["result_arena_placement"]["probe_GBps_of_each_candidate"])
P
done
for i in 1 2; do
  RHIP_ARENA_VMM=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary > $O/bench_old_$i.json 2> $O/bench_old_$i.err
  python - $O/bench_old_$i.json <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print("old", round(d["value"]), round(d["roofline"]["frac"], 4), d["config"]["c2_fresh_result_pool_ms"]["and"], d["config"]["c2_fresh_result_pool_ms"]["or"], d["config"]["result_arena_placement"]["probe_GBps_of_each_candidate"])
P
done
