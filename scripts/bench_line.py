"""bench.py's JSON line (file argument or stdin) -> a few short lines: the headline, then one row per secondary."""
import json, sys
src = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads([l for l in src.strip().split("\n") if l.startswith("{")][-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "k_bb ms", round(d["roofline"]["avg_launch_ms"], 3),
      "frac", round(d["roofline"]["frac"], 4), "transport", d.get("transport"))
c = d["config"]
print("fresh pool ms", c.get("c2_fresh_result_pool_ms", {}).get("and"), c.get("c2_fresh_result_pool_ms", {}).get("or"),
      "steady", c.get("c2_fresh_result_pool_ms", {}).get("steady_and"), c.get("c2_fresh_result_pool_ms", {}).get("steady_or"),
      "probes", c.get("result_arena_placement", {}).get("probe_GBps_of_each_candidate"))
for k, r in c.get("secondary_summary", {}).get("rows", {}).items():
    print(" ", k, r)
print("cpu_baseline", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k in ("value", "cores", "kind", "one_core_ops_per_s", "error")})
