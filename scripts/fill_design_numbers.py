"""Regenerate the measured-numbers block of DESIGN.md (between the NUMBERS markers) from profiles/r03_*."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(ROOT, "profiles", n)
d = json.load(open(P("r03_bench.json")))
s = d["config"]["secondary"]; r = d["roofline"]; cb = d["cpu_baseline"]
q = {x["dataset"]: x for x in map(json.loads, open(P("r03_quick_c3.jsonl")))}
traffic = json.load(open(P("bb_traffic.json")))


def row(k, label):
    v = s[k]; cpu = v["cpu1_ops_per_s"]
    pipe = f"{v['ms_batch_pipelined2']:.3f}" if "ms_batch_pipelined2" in v else "—"
    return (f"| {label} | {v['pairs']:,} | {v['ms_batch_median']:.3f} | {pipe} | {v['ops_per_s'] / 1e6:.1f} M | {v['alg_GBps'] / 1e3:.2f} | "
            f"{cpu / 1e3:,.0f} k | {v['ops_per_s'] / cpu:,.0f}× | {'ok' if v['checksum_ok'] else 'FAIL'} |")


t = []
t.append(f"""**Headline (`bench.py`, C2, N = 1, driver contract).** {d['value']:,.0f} set-ops/s = {d['config']['algorithmic_GBps'] / 1e3:.2f} TB/s algorithmic over a
{d['config']['timed_region_s']:.2f} s timed region ({d['ms_per_step']:.1f} ms per step of 3 000 ops).  Dominant kernel `k_bb`: {r['achieved'] / 1e3:.2f} TB/s = **{r['frac']:.3f} of the 8 TB/s
HBM peak** (average launch {r['avg_launch_ms']:.2f} ms over {r['launches_timed']} launches timed with HIP events on the engine's stream; 1 024 000 container
pairs × 24 576 B per launch); HBM traffic from the PMC passes of the same measurement pass = {traffic['hbm_bytes_per_launch'] / 1e9:.2f} GB per launch =
**{traffic['hbm_bytes_per_launch'] / traffic['algorithmic_bytes']:.4f} × algorithmic** (`profiles/r03_pmc_summary.md`: FETCH_SIZE ×2 per the gfx950 correction, calibrated on
`k_synth_dir` / `k_synth_fill`).  `rocprofv3 --kernel-trace --stats` of the same command: `profiles/r03_bench_c2_kernel_stats.csv`.
CPU baseline = the real CRoaring (`oracle/_ref`, AVX-512 build) on the box's host ({cb['host_threads']} hardware threads): {cb['one_core']['ops_per_s_median']:.0f} ops/s on one core
({cb['one_core']['GBps_median']:.1f} GB/s); worker-process sweep {', '.join(f"{T}: {v['ops_per_s_median']:,.0f}" for T, v in cb['sweep'].items())} ops/s — best {cb['value']:,.0f} at {cb['cores']} processes,
falling from there (the host's memory system, not its core count, is the limit: every op allocates and writes a 32 MiB result);
ISA variants of one core: AVX2 {cb['isa_1core']['avx2']['ops_per_s_median_1core']:.0f}, AVX-512 {cb['isa_1core']['avx512']['ops_per_s_median_1core']:.0f}, scalar
{cb['isa_1core']['scalar']['ops_per_s_median_1core']:.0f} ops/s.  GPU / best CPU = {d['value'] / cb['value']:.0f}×; the number that says something about the kernel is the {r['frac']:.2f}.
""")
t.append("Per op on C2 (250 pairs per call, one result pool per op, three consecutive runs of `scripts/bench_c2_ops.py`: `profiles/r03_c2_ops.jsonl`; ms per call / `k_bb` ms / `k_bb` TB/s):\n\n| op | run 1 | run 2 | run 3 |\n|---|---|---|---|")
runs = {}
for o in map(json.loads, open(P("r03_c2_ops.jsonl"))):
    runs.setdefault(o["op"], []).append(o)
for op, rs in runs.items():
    t.append(f"| {op} | " + " | ".join(f"{o['ms_call']:.2f} / {o['k_bb_ms']:.3f} / {o['k_bb_GBps'] / 1e3:.2f}" for o in rs) + " |")
t.append("""
(`k_bb<and>` against `k_bb<or>`: within 0.5 % in all three runs.  The "bimodality" of rounds 1-2 is WHERE THE RESULT ARENA
LANDS: with a freshly allocated result pool per op `k_bb<xor>` takes 4.63 ms and `k_bb<or>` 4.40 ms, recycling the other
op's arena swaps them (`scripts/c2_or_clock.py`) -- the three streams of the kernel advance in lockstep, so their relative
placement holds for the whole launch.  Run down in the second half of round 3 (`scripts/arena_skew_sweep.py`): ONE
pool, ONE process, the result arena freed and re-allocated for every row -- the same virtual address gives 4.37-4.45 ms
or 4.64-4.71 ms at random, whatever the arena's offset into its allocation (256 B ... 1 GiB) and whatever its size is
rounded to (1 ... 16 GiB).  It is the PHYSICAL pages the driver hands out, nothing the address shows, and consecutive
allocations come in streaks of one mode (eight slow ones in a row have been seen).  So the only handle is to look:
`Engine.pairwise_placed` runs a batch into `tries` freshly allocated result pools, all kept alive until the end, and
keeps the fastest `keep` of them (a result pool serves any op); `bench.py` does that once at start-up, untimed, for the
two result pools its steps recycle (`config.result_arena_startup` lists the `k_bb` time of every try; `--arena-tries 0`
takes the first allocations).  Twelve candidates, the best two kept: four consecutive bench runs gave 0.713 / 0.717 /
0.718 / 0.722 of peak; with eight tries per op, separately, one run of four still had eight slow `and` arenas in a row
(0.692).  Cardinality mode: the events of the call's own slot are read now; round 2's
table repeated a stale pair.)

**Realdata, ALL unordered pairs in one batched call per op** (`bench.py` secondary block: wall time of the whole call
incl. planning and the final wait, median of >= 10 calls; "2 in flight" = per-call period of 40 calls issued with
`rhip_pairwise_begin` / `_end`, two at a time; checksum = sum of result cardinalities against the reference fixture; CPU =
real CRoaring, one core, same pairs):

| config | pairs | ms / batch | ms, 2 in flight | set-ops/s | alg. TB/s | CRoaring 1 core | ratio | checksum |
|---|---|---|---|---|---|---|---|---|""")
for k, l in (("c3_and", "C3 weather_sept_85 and"), ("c3_or", "C3 or"), ("c3_xor", "C3 xor"), ("c3_andnot", "C3 andnot"),
             ("c1_and", "C1 census1881 and"), ("c1_or", "C1 or"), ("c1_xor", "C1 xor"), ("c1_andnot", "C1 andnot"),
             ("c5_and", "C5 roaring64 wikileaks×10 and"), ("c5_or", "C5 or")):
    t.append(row(k, l))
c4 = s["c4_or_many"]; u = s["c5_union_200"]
w = q["wikileaks-noquotes"]; ci = q["census-income"]
pp = lambda x, o: f"{x[o]['ms']:.3f}" + (f" ({x[o]['ms_pipelined2']:.3f})" if "ms_pipelined2" in x[o] else "")
for k, l in (("c3_multi4", "C3 and+or+xor+andnot, ONE batch (`rhip_pairwise_multi`)"), ("c1_multi4", "C1 four ops, one batch"), ("c5_multi2", "C5 and+or, one batch")):
    v = s[k]
    t.append(f"| {l} | {v['pairs']:,} x {len(v['ops'])} | {v['ms_batch_median']:.3f} (as separate calls: {v['ms_sum_of_single_op_batches']:.3f}) | -- | {v['ops_per_s'] / 1e6:.1f} M | {v['alg_GBps'] / 1e3:.2f} | -- | -- | {'ok' if v['checksum_ok'] else 'FAIL'} |")
t.append(f"""| C4 `or_many`, 100 000 sparse bitmaps (3.2 M containers) | -- | {c4['ms_median']:.2f} | -- | {c4['ops_per_s']:.0f} | {c4['alg_GBps'] / 1e3:.2f} | {c4['cpu1_ms_first_10000']:.0f} ms for the first 10 000 | ~ {c4['cpu1_ms_first_10000'] * 10 / c4['ms_median']:,.0f}x | cardinality ok |
| C4 through the SHARDED pipeline on a one-rank group (stage 1 -> dense table -> stage 3, one wait) | -- | {c4['sharded_w1']['ms_median']:.2f} ({c4['sharded_w1']['vs_or_many']:.2f} x `or_many`); with the all-to-all issued on a 1-rank nccl group {c4['sharded_w1_nccl']['ms_median']:.2f} ({c4['sharded_w1_nccl']['vs_or_many']:.2f} x) | -- | -- | -- | -- | -- | cardinality ok |
| C5 union of 200 roaring64 bitmaps | -- | {u['ms_median']:.2f} | -- | -- | -- | {u['cpu1_ms_fold']:.1f} ms (fold) | {u['cpu1_ms_fold'] / u['ms_median']:.0f}x | cardinality ok |

Cardinality-only batches: C3 `and` {s['c3_and_cardinality']['ms_batch_median']:.3f} ms, C1 {s['c1_and_cardinality']['ms_batch_median']:.3f} ms, C5 {s['c5_and_cardinality']['ms_batch_median']:.3f} ms.  Other sets of the corpus
(`profiles/r03_quick_c3.jsonl`, min of 7, in brackets the period with two calls in flight): census-income and / or / xor /
andnot {pp(ci, 'and')} / {pp(ci, 'or')} / {pp(ci, 'xor')} / {pp(ci, 'andnot')} ms, wikileaks-noquotes {pp(w, 'and')} / {pp(w, 'or')} /
{pp(w, 'xor')} / {pp(w, 'andnot')} ms.  All four ops in one batch (`profiles/r03_multi_ops.txt`): census-income 0.97 ms against 1.40,
wikileaks 0.53 against 0.83.

Round 2 -> round 3, ms per batch (round 2's driver line -> this pass; different boxes, and boxes differ by +-5 % on these
small batches): weather `and` 0.361 -> {s['c3_and']['ms_batch_median']:.3f}, `or` 0.769 -> {s['c3_or']['ms_batch_median']:.3f}, `xor` 0.772 -> {s['c3_xor']['ms_batch_median']:.3f}, `andnot` 0.561 -> {s['c3_andnot']['ms_batch_median']:.3f};
census1881 `and` 0.162 -> {s['c1_and']['ms_batch_median']:.3f} ({s['c1_and'].get('ms_batch_pipelined2', float('nan')):.3f} with two calls in flight), `or` 0.334 -> {s['c1_or']['ms_batch_median']:.3f}; C5 `and` 0.574 -> {s['c5_and']['ms_batch_median']:.3f}, `or` 0.984 -> {s['c5_or']['ms_batch_median']:.3f};
C4 `or_many` 1.69 -> {c4['ms_median']:.2f}; C5 union 0.27 -> {u['ms_median']:.2f}.  Same-box A/B of the round's three pairwise changes (min of 7 calls,
`scripts/gpu_r3b.sh` / `_r3f.sh` / `_r3h.sh`): `k_ba` weather `andnot` 0.560 -> 0.512; one-wave `k_genw` weather `and` 0.362 -> 0.338;
staged `k_usmall` output weather `or` / `xor` 0.757 / 0.770 -> 0.725 / 0.748.  The many-way path and the multi-op batch are where
this round's factors are; the single-op realdata fractions of the HBM peak stay at {s['c3_and']['frac']:.2f} / {s['c3_or']['frac']:.2f} / {s['c3_xor']['frac']:.2f} / {s['c3_andnot']['frac']:.2f} (weather)
and {s['c1_and']['frac']:.2f}-{s['c1_or']['frac']:.2f} (census1881): section 8 says what bounds them.

**Per-kernel algorithmic GB/s on weather** (`profiles/r03_per_kernel_c3.jsonl`: `rhip_last_class_stats` joined with the
kernels' STAND-ALONE durations -- one stream, `RHIP_NO_OVERLAP=1`; operands are L2-resident, so figures above the HBM peak are
cache bandwidth):

| op | kernel | container pairs | MB in / out | us alone | alg. GB/s |
|---|---|---|---|---|---|""")
for l in open(P("r03_per_kernel_c3.jsonl")):
    x = json.loads(l)
    if x["items"] >= 5000:
        t.append(f"| {x['op']} | `{x['kernel']}` | {x['items']:,} | {x['MB_in']:.0f} / {x['MB_out']:.0f} | {x['us_standalone']:.0f} | {x['alg_GBps']:,.0f} |")
t.append("""
**Class throughput** (`profiles/r03_class_throughput.jsonl`; 147 456 container pairs of one type pair per batch, 25–50 MB
of operands — cache-resident, so the TB/s of the bitset rows exceed what HBM gives; ns per container pair is the
comparable figure):

| pair | and | or | xor | andnot |
|---|---|---|---|---|""")
for l in open(P("r03_class_throughput.jsonl")):
    x = json.loads(l)
    if "pair" in x:
        t.append(f"| {x['pair']} | " + " | ".join(f"{x[o]['ns_per_item']:.2f} ns, {x[o]['TBps']:.2f} TB/s" for o in ("and", "or", "xor", "andnot")) + " |")
t.append("""
(round 2 -> 3: A874 x B / B x A874 `or`, `xor` and B x A874 `andnot` now run through `k_ba`.)  SQ counters of the class
kernels on weather: `profiles/r03_pmc_weather_sq.md` (re-taken: the LDS bank-conflict share of the image kernels is what a
RANDOM scatter gives -- 32 lanes into 32 banks put ~3.5 on the fullest bank -- and a swizzle moves addresses, not the
collision statistics: `k_filter` 42 %, `k_wave` 33 %, `k_many_l1` 57 % with its XOR swizzle in place); of the many-way
kernels on C4: `profiles/r03_pmc_c4_sq.md`.  Kernel timelines of one batch of every configuration:
`profiles/r03_timelines.txt`; per-kernel `--stats` tables: `profiles/r03_{c1,c3,c5,wk,c4,multi_c1}_*_kernel_stats.csv`.""")
body = "\n".join(t)
p = os.path.join(ROOT, "DESIGN.md")
txt = open(p).read()
a = txt.index("<!-- NUMBERS:BEGIN"); a = txt.index("\n", a) + 1
b = txt.index("<!-- NUMBERS:END -->")
open(p, "w").write(txt[:a] + body + "\n" + txt[b:])
print("DESIGN.md numbers block regenerated")
