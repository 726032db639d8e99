"""Regenerate the measured-numbers block of DESIGN.md (between the NUMBERS markers) from profiles/r06_* (the files
scripts/collect_profiles.py r06 copies out of the round-6 measurement pass, scripts/gpu_final_r6.sh).  Fails loudly on a
missing or empty input."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r06"


def P(n):
    p = os.path.join(ROOT, "profiles", n)
    if not os.path.exists(p) or os.path.getsize(p) == 0:
        raise SystemExit(f"fill_design_numbers: {p} is missing or empty")
    return p


d = json.load(open(P(f"{TAG}_bench.json")))
s = json.load(open(P(f"{TAG}_bench_detail.json")))["detail"]
r = d["roofline"]
cb = s["cpu_baseline_full"]
traffic = json.load(open(P("bb_traffic.json")))
qa = {}
for line in open(P(f"{TAG}_quick_all.txt")):
    m = re.match(r"(\w+) (\S+) (\{.*\})", line)
    if m:
        qa.setdefault(m.group(1), {})[m.group(2)] = json.loads(m.group(3))
    m = re.match(r"(\w+) (\S+) multi4 ms ([\d.]+) sum of singles ([\d.]+)", line)
    if m:
        qa.setdefault(m.group(1) + "_multi", {})[m.group(2)] = (float(m.group(3)), float(m.group(4)))


def row(k, label):
    v = s[k]
    cpu = v.get("cpu1_ops_per_s")
    return (f"| {label} | {v['pairs']:,} | {v['ms_batch_median']:.3f} ({v['ms_batch_min']:.3f}) | {v.get('ms_first_call', float('nan')):.3f} | {v['ms_adhoc_list']:.3f} | {v['ms_batch_pipelined2']:.3f} | "
            f"{v['ops_per_s'] / 1e6:.1f} M | {v['alg_GBps'] / 1e3:.2f} | **{v['frac']:.3f}** | {('%.2f' % v['hbm_traffic_frac']) if 'hbm_traffic_frac' in v else '--'} | {cpu / 1e3:,.0f} k | {v['ops_per_s'] / cpu:,.0f}× | {'ok' if v['checksum_ok'] else 'FAIL'} |")


pl = d["config"]["result_arena_placement"]["probe_GBps_of_each_candidate"]
FRESH = d['config'].get('c2_fresh_result_pool_ms', {})
t = []
t.append(f"""**Headline (`bench.py`, C2, N = 1, driver contract).** {d['value']:,.0f} set-ops/s = {d['config']['algorithmic_GBps'] / 1e3:.2f} TB/s algorithmic over a
{d['config']['timed_region_s']:.2f} s timed region ({d['ms_per_step']:.1f} ms per step of 3 000 ops).  Dominant kernel `k_bb`: {r['achieved'] / 1e3:.2f} TB/s = **{r['frac']:.3f} of the 8 TB/s
HBM peak** (average launch {r['avg_launch_ms']:.2f} ms over {r['launches_timed']} launches timed with HIP events on the engine's stream; 1 024 000 container
pairs × 24 576 B per launch); HBM traffic MEASURED BY THE BENCH RUN ITSELF (round 6: two `rocprofv3 --pmc` child runs, FETCH_SIZE and
WRITE_SIZE in separate passes, `live_bb_traffic` in `bench.py`) = {r['traffic'] / 1e9:.2f} GB per launch = **{r['traffic'] / (r['pairs_per_launch'] * 24576):.4f} × algorithmic**; the
PMC passes of `scripts/gpu_final_r6.sh` on the same box: {traffic['hbm_bytes_per_launch'] / 1e9:.2f} GB = {traffic['hbm_bytes_per_launch'] / traffic['algorithmic_bytes']:.4f} × (`profiles/{TAG}_pmc_summary.md`: FETCH_SIZE ×2 per the gfx950
correction, calibrated on `k_synth_dir` / `k_synth_fill`).  `rocprofv3 --kernel-trace --stats` of the same command: `profiles/{TAG}_bench_c2_kernel_stats.csv`.
The two result arenas were placed by the library itself (`place_arena`, §3): probe rates of the candidate allocations and then of the address positions visited, GB/s --
`and`: {pl.get('and')}, `or`: {pl.get('or')}.  Fresh processes on the same box (`profiles/{TAG}_placement_runs.txt`): see below.
CPU baseline = the real CRoaring (`oracle/_ref`, AVX-512 build) on the box's host ({cb['host_threads']} hardware threads): {cb['one_core']['ops_per_s_median']:.0f} ops/s on one core
({cb['one_core']['GBps_median']:.1f} GB/s); worker-process sweep {', '.join(f"{T}: {v['ops_per_s_median']:,.0f}" for T, v in cb['sweep'].items())} ops/s — best {cb['value']:,.0f} at {cb['cores']} processes,
falling from there: CRoaring is **malloc-bound** on this workload (every op allocates, page-faults and writes a 32 MiB result; the
reference has no result-reuse API), so the host's memory system, not its core count, is the limit.  ISA variants of one core:
AVX2 {cb['isa_1core']['avx2']['ops_per_s_median_1core']:.0f}, AVX-512 {cb['isa_1core']['avx512']['ops_per_s_median_1core']:.0f}, scalar {cb['isa_1core']['scalar']['ops_per_s_median_1core']:.0f} ops/s.  GPU / best CPU = {d['value'] / cb['value']:.0f}×; the number that says
something about the kernel is the {r['frac']:.2f}.
""")
t.append(f"Per op on C2 (250 pairs per call, one result pool per op, each arena placed by the library; `scripts/bench_c2_ops.py`: `profiles/{TAG}_c2_ops.jsonl`):\n\n"
         "| op | ms per call | `k_bb` ms | `k_bb` TB/s | of peak |\n|---|---|---|---|---|")
for o in map(json.loads, open(P(f"{TAG}_c2_ops.jsonl"))):
    t.append(f"| {o['op']} | {o['ms_call']:.2f} | {o['k_bb_ms']:.3f} | {o['k_bb_GBps'] / 1e3:.2f} | {o['k_bb_GBps'] / 8000:.3f} |")
t.append(f"""
(The cardinality forms write nothing: 16 384 B per pair.  WHERE IN THE ADDRESS SPACE an arena sits moves `k_bb` between 3.9 and
4.7 ms -- §3: its physical pages on some boxes, its virtual address on others; the library measures in both.  A call that has to ALLOCATE its result pool
-- no `reuse` -- took {FRESH.get('and')} ms (`and`) / {FRESH.get('or')} ms (`or`) in this pass, address search included; a caller that frees its
result and calls again gets the parked arena back: {FRESH.get('steady_and', float('nan')):.2f} / {FRESH.get('steady_or', float('nan')):.2f} ms per call (`c2_fresh_result_pool_ms` in the bench line).)

Fresh processes, one after the other on the box of this pass (`profiles/{TAG}_placement_runs.txt`; value, `k_bb` ms, fraction, fresh-pool
ms, positions probed):

```
{open(P(f"{TAG}_placement_runs.txt")).read().strip()}
```

**Realdata, ALL unordered pairs in one batched call per op** (`bench.py`, detail in `profiles/{TAG}_bench_detail.json`: wall time
of the whole call incl. the final wait over a PREPARED pair list (`rhip_pairlist_all_pairs`) whose PLAN IS KEPT WITH THE LIST
(round 6: a repeated batch starts at its class kernels), median (min) of >= 10 calls; "first call" = the same call when the
list holds no plan yet (`rhip_pairlist_drop_plans` before it: planning kernels included -- round 5's figure); "ad hoc" = the
same call handed the two index arrays per call (plans every time); "2 in flight" = per-call period of 40 calls issued with
`rhip_pairwise_list_begin` / `_end`, two at a time; checksum = sum of result cardinalities against the reference fixture;
CPU = real CRoaring, one core, same pairs; "HBM traffic" = bytes that crossed the memory side per batch in the stored PMC
pass (`profiles/{TAG}_realdata_traffic.md`) over the batch time, as a fraction of 8 TB/s: the operands of these sets live in
the L2s, so the algorithmic fraction is NOT HBM utilisation -- the bound of every row is instruction issue + LDS + latency,
with the results streaming out).  The same calls by `scripts/quick_all.py`, min of 7:
weather {qa['prepared']['weather_sept_85']}, census1881 {qa['prepared']['census1881']}.

| config | pairs | ms / batch, median (min) | first call | ad hoc list | 2 in flight | set-ops/s | alg. TB/s | of HBM peak (algorithmic) | HBM traffic / peak | CRoaring 1 core | ratio | checksum |
|---|---|---|---|---|---|---|---|---|---|---|---|---|""")
for k, l in (("c3_and", "C3 weather_sept_85 and"), ("c3_or", "C3 or"), ("c3_xor", "C3 xor"), ("c3_andnot", "C3 andnot"),
             ("c1_and", "C1 census1881 and"), ("c1_or", "C1 or"), ("c1_xor", "C1 xor"), ("c1_andnot", "C1 andnot"),
             ("c5_and", "C5 roaring64 wikileaks×10 and"), ("c5_or", "C5 or")):
    t.append(row(k, l))
c4, x10, u, st = s["c4_or_many"], s["c4x10_or_many"], s["c5_union_200"], s["c4_shard_stages"]
t.append("")
t.append("| config | ms | alg. TB/s | of HBM peak | note |\n|---|---|---|---|---|")
for k, l in (("c3_multi4", "C3 and+or+xor+andnot, ONE batch"), ("c1_multi4", "C1 four ops, one batch"), ("c5_multi2", "C5 and+or, one batch")):
    v = s[k]
    t.append(f"| {l} | {v['ms_batch_median']:.3f} | {v['alg_GBps'] / 1e3:.2f} | {v['frac']:.3f} | as separate calls {v['ms_sum_of_single_op_batches']:.3f} ms; checksum {'ok' if v['checksum_ok'] else 'FAIL'} |")
t.append(f"| C4 `or_many`, 100 000 sparse bitmaps (3.2 M containers, 1.64 GB) | {c4['ms_median']:.3f} | {c4['alg_GBps'] / 1e3:.2f} | {c4['frac']:.3f} | CRoaring 1 core: {c4['cpu1_ms_first_10000']:.0f} ms for the first 10 000; sharded pipeline at world 1: {c4['sharded_w1']['vs_or_many']:.2f} x, with the collective on a 1-rank nccl group {c4['sharded_w1_nccl']['vs_or_many']:.2f} x; cardinality ok |")
t.append(f"| C4 x 10 `or_many`, 10^6 sparse bitmaps (32 M containers, 16.4 GB) | {x10['ms_median']:.2f} | {x10['alg_GBps'] / 1e3:.2f} | {x10['frac']:.3f} | pool built in {x10['build_s_untimed']:.1f} s (untimed); cardinality equal to the reference's (`tests/golden/c4x10_or_many.npz`) |")
t.append(f"| C5 union of 200 roaring64 bitmaps | {u['ms_median']:.3f} | -- | -- | CRoaring fold {u['cpu1_ms_fold']:.1f} ms; cardinality ok |")
t.append("")
t.append("**The reference benchmark's own loop shape** (`benchmarks/benchmark.cpp:2035-2091` successive_and / successive_or: the n - 1 "
         "adjacent pairs, each result materialised + cardinality; here ONE batch over `rhip_pairlist_successive` + the cardinalities "
         "read back -- almost pure fixed cost of a call) and the per-call drop-in:\n\n| row | pairs | ms / batch | us per op | CRoaring 1 core, us per op | GPU / CPU |\n|---|---|---|---|---|---|")
for k in ("c1_successive_and", "c1_successive_or", "c3_successive_and", "c3_successive_or"):
    if k in s:
        v = s[k]
        t.append(f"| {k} | {v['pairs']} | {v['ms_batch_median']:.3f} | {v['us_per_op']:.3f} | {v.get('cpu1_us_per_op', float('nan')):.3f} | {v.get('gpu_over_cpu1', float('nan')):.2f} x |")
if "dropin_percall_us" in s and "and" in s["dropin_percall_us"]:
    dp = s["dropin_percall_us"]
    t.append(f"| per-call drop-in `roaring_bitmap_and` / `_or` on census1881 operands (host struct -> upload -> batch of one -> download) | 1 | -- | {dp['and']['dropin_us_per_call']:.0f} / {dp['or']['dropin_us_per_call']:.0f} | {dp['and']['croaring_us_per_call']:.2f} / {dp['or']['croaring_us_per_call']:.2f} | {1 / dp['and']['dropin_over_cpu']:.4f} x / {1 / dp['or']['dropin_over_cpu']:.4f} x |")
t.append(f"""
Cardinality-only batches: C3 `and` {s['c3_and_cardinality']['ms_batch_median']:.3f} ms, C1 {s['c1_and_cardinality']['ms_batch_median']:.3f} ms, C5 {s['c5_and_cardinality']['ms_batch_median']:.3f} ms.

**One rank of N, measured on one GPU** (`bench.py` `c4_shard_stages`: rank 0's own pool -- bitmaps 0, N, 2N ... -- stage 1
`rhip_many_partials_dense` into a world = N table and stage 3 `rhip_many_finalize_dense` over a world = N table, each timed to
completion; beside them round 3's model `0.10 + 0.69 / N` and `0.03`, which the counting-sort grouping of round 5 now beats):

| N | stage 1 ms | stage 3 ms | model stage 1 | model stage 3 |
|---|---|---|---|---|""")
for n in ("1", "2", "4", "8"):
    v = st[n]
    t.append(f"| {n} | {v['stage1_ms']:.3f} | {v['stage3_ms']:.3f} | {v['model_stage1_ms']:.3f} | {v['model_stage3_ms']:.3f} |")
t.append(f"""
(§7a: round 4 measured 0.759 / 0.470 / 0.440 / 0.319 ms for stage 1 -- a fixed part of ~0.25 ms; it is ~0.08 ms now.)

**The corpus, back to back** (`profiles/{TAG}_quick_all.txt`: all pairs, min of 7 synchronous calls, ms; first the prepared pair
list, then the ad hoc one):

| data set | and | or | xor | andnot | four ops in one batch (sum of the four) |
|---|---|---|---|---|---|""")
for ds in ("weather_sept_85", "census1881", "census-income", "wikileaks-noquotes", "c5"):
    a, b = qa["prepared"][ds], qa["adhoc"][ds]
    m = qa.get("prepared_multi", {}).get(ds)
    t.append(f"| {ds} | {a['and']:.3f} / {b['and']:.3f} | {a['or']:.3f} / {b['or']:.3f} | {a['xor']:.3f} / {b['xor']:.3f} | {a['andnot']:.3f} / {b['andnot']:.3f} | " +
             (f"{m[0]:.3f} ({m[1]:.3f})" if m else "--") + " |")
t.append(f"""
**Per-kernel algorithmic GB/s on weather** (`profiles/{TAG}_per_kernel_c3.jsonl`: `rhip_last_class_stats` joined with the
kernels' STAND-ALONE durations -- one stream, `RHIP_NO_OVERLAP=1`; operands are L2-resident, so figures above the HBM peak are
cache bandwidth; an X-grouped batch serves `k_wave` and `k_ba` items with ONE `k_union_g` launch, whose time both rows show):

| op | kernel | container pairs | MB in / out | us alone | alg. GB/s |
|---|---|---|---|---|---|""")
for l in open(P(f"{TAG}_per_kernel_c3.jsonl")):
    x = json.loads(l)
    if x["items"] >= 5000 and x["us_standalone"]:
        t.append(f"| {x['op']} | `{x['kernel']}` | {x['items']:,} | {x['MB_in']:.0f} / {x['MB_out']:.0f} | {x['us_standalone']:.0f} | {x['alg_GBps']:,.0f} |")
t.append(f"""
**Class throughput** (`profiles/{TAG}_class_throughput.jsonl`; 147 456 container pairs of one type pair per batch, 25–50 MB
of operands — cache-resident, so the TB/s of the bitset rows exceed what HBM gives; ns per container pair is the
comparable figure):

| pair | and | or | xor | andnot |
|---|---|---|---|---|""")
for l in open(P(f"{TAG}_class_throughput.jsonl")):
    x = json.loads(l)
    if "pair" in x:
        t.append(f"| {x['pair']} | " + " | ".join(f"{x[o]['ns_per_item']:.2f} ns, {x[o]['TBps']:.2f} TB/s" for o in ("and", "or", "xor", "andnot")) + " |")
t.append(f"""
SQ counters of the class kernels on weather: `profiles/{TAG}_pmc_weather_sq.md` (taken with kernels serialised by the
profiler -- the context's self-test then joins with events, §2); L2 hit / miss and memory requests per class kernel:
`profiles/r04_pmc_l2.md`; HBM bytes per batch and issue shares of every realdata row: `profiles/{TAG}_realdata_traffic.md`.
Kernel timelines of one batch of every configuration: `profiles/{TAG}_timelines.txt`, of the many-way calls (C4 and C4 x 10):
`profiles/{TAG}_timelines_many.txt` -- six launches of the engine's own, no library kernel, no fill or copy command;
SQ counters of the many-way kernels before / after the round: `profiles/{TAG}_pmc_c4_sq_before.md`, `profiles/{TAG}_pmc_c4_sq.md`;
per-kernel `--stats` tables: `profiles/{TAG}_{{c1,c3,c5,c4,c4x10}}_*_kernel_stats.csv`.""")
body = "\n".join(t)
p = os.path.join(ROOT, "DESIGN.md")
txt = open(p).read()
a = txt.index("<!-- NUMBERS:BEGIN")
a = txt.index("\n", a) + 1
b = txt.index("<!-- NUMBERS:END -->", a)
open(p, "w").write(txt[:a] + body + "\n" + txt[b:])
print("DESIGN.md numbers block regenerated")
