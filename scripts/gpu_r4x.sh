cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4x
timeout 55 python bench.py --no-cpu --steps 10 --warmup 2 > gpurun_out/r4x/bench.json 2> gpurun_out/r4x/bench.err; echo "rc=$?"; wc -c gpurun_out/r4x/bench.json
python - <<'P'
import json
d = json.load(open("gpurun_out/r4x/bench.json"))
r = d["config"]["secondary_summary"]["rows"]
print(round(d["value"]), round(d["roofline"]["frac"], 4), {k: r[k] for k in ("c3_and", "c3_or", "c3_xor", "c3_andnot", "c1_and", "c5_and", "c5_or", "c4_or_many", "c4x10_or_many") if k in r})
P
grep -v BENCH_DETAIL gpurun_out/r4x/bench.err | tail -3
