"""The PMC passes of scripts/gpu_final_r5.sh over the realdata batches (rocprofv3 --pmc on scripts/prof_weather.py, 12
batches per run; FETCH_SIZE, WRITE_SIZE and an SQ set in SEPARATE runs) -> profiles/realdata_traffic.json, which bench.py
reads for the `hbm_traffic_frac` / `bound` columns of its realdata rows, and profiles/<tag>_realdata_traffic.md.
FETCH_SIZE is doubled (the gfx950 half-count of wide reads, MI355X_MICROARCH.md HBM; calibrated on k_synth_dir in
profiles/r04_pmc_summary.md), WRITE_SIZE taken as it is (calibrated 1.0000 on k_synth_fill there).  Unit: KiB.
Usage: python scripts/summarize_realdata_traffic.py <dir of the pass> <tag>"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r05")
N_BATCH = 12  # scripts/prof_weather.py
SKIP = ("k_des_", "k_bitmap_bounds", "k_key_", "k_conc_probe", "k_payload_stats", "k_pairlist", "k_synth", "rocclr", "k_join_signal")

def counters(d):
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    if not fs:
        return agg
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(s in k for s in SKIP):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    return agg

out, lines = {}, [f"# HBM traffic and issue shares of the realdata batches ({tag})", "",
                  "`rocprofv3 --pmc <set> --kernel-trace` on `scripts/prof_weather.py <op> <set>` (12 all-pairs batches over a prepared pair",
                  "list; FETCH_SIZE, WRITE_SIZE and the SQ set each in a run of its own; kernels run one at a time under the counters).",
                  "read = FETCH_SIZE x 2 x 1024 (gfx950 half-count), write = WRITE_SIZE x 1024, per batch; algorithmic bytes from the",
                  "bench line.  `valu` = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, `wait` = SQ_WAIT_ANY / SQ_WAVE_CYCLES, `lds conflict` =",
                  "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, over the batch's kernels.", "",
                  "| row | read MB / batch | write MB / batch | valu | wait | lds conflict | heaviest kernels (read + write MB) |", "|---|---|---|---|---|---|---|"]
for d in sorted(glob.glob(os.path.join(src, "rt_*_fetch"))):
    row = os.path.basename(d)[3:-6]
    F, W, S = counters(d), counters(d.replace("_fetch", "_write")), counters(d.replace("_fetch", "_sq"))
    rd = sum(v.get("FETCH_SIZE", 0.0) for v in F.values()) * 2 * 1024 / N_BATCH
    wr = sum(v.get("WRITE_SIZE", 0.0) for v in W.values()) * 1024 / N_BATCH
    wc = sum(v.get("SQ_WAVE_CYCLES", 0.0) for v in S.values())
    valu = sum(v.get("SQ_ACTIVE_INST_VALU", 0.0) for v in S.values()) / wc if wc else None
    wait = sum(v.get("SQ_WAIT_ANY", 0.0) for v in S.values()) / wc if wc else None
    la = sum(v.get("SQ_LDS_IDX_ACTIVE", 0.0) for v in S.values())
    conf = sum(v.get("SQ_LDS_BANK_CONFLICT", 0.0) for v in S.values()) / la if la else None
    per = {k: (F.get(k, {}).get("FETCH_SIZE", 0.0) * 2 * 1024 + W.get(k, {}).get("WRITE_SIZE", 0.0) * 1024) / N_BATCH / 1e6 for k in set(F) | set(W)}
    top = ", ".join(f"{k} {v:.0f}" for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:4])
    out[row] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "valu_share_of_wave_cycles": valu, "wait_share_of_wave_cycles": wait,
                "lds_bank_conflict_share": conf, "source": f"profiles/{tag}_realdata_traffic.md"}
    f = lambda x: "-" if x is None else f"{100 * x:.0f} %"
    lines.append(f"| {row} | {rd / 1e6:.1f} | {wr / 1e6:.1f} | {f(valu)} | {f(wait)} | {f(conf)} | {top} |")
if not out:
    sys.exit("no rt_*_fetch directories under " + src)
json.dump(out, open(os.path.join(ROOT, "profiles", "realdata_traffic.json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", f"{tag}_realdata_traffic.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
