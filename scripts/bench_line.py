"""stdin: bench.py's JSON line -> one short line (value, ms per step, k_bb ms, frac, start-up tries)."""
import json, sys
d = json.loads(sys.stdin.read().strip().split("\n")[-1])
print(round(d["value"]), round(d["ms_per_step"], 2), round(d["roofline"]["avg_launch_ms"], 3), round(d["roofline"]["frac"], 4),
      d["config"].get("result_arena_placement", d["config"].get("result_arena_startup", {})).get("probe_GBps_of_each_candidate"),
      d["config"].get("result_arena_placement", d["config"].get("result_arena_startup", {})).get("k_bb_ms_of_each_try"))
