"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide
prescribes) into profiles/r01_pmc_summary.md and profiles/bb_traffic.json (read by bench.py)."""
import collections, csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"

def load(d, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(src, d, "b_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return agg

F, W = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
KiB = 1024.0
# calibration kernels with exactly known traffic (8 GiB pool): k_synth_fill writes it once, k_synth_dir reads it once
cal_w = W["k_synth_fill"][0] * KiB / (8 * 2**30)
cal_r = F["k_synth_dir"][0] * KiB / (8 * 2**30)
lines = [f"# rocprofv3 PMC summary ({tag}) -- bench.py C2 workload, 1 MI355X", "",
         "Counters collected in separate passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`), unit = KiB.",
         f"Calibration on known byte counts: k_synth_fill writes 8 GiB -> WRITE_SIZE x1024 / 8 GiB = {cal_w:.4f};",
         f"k_synth_dir reads 8 GiB -> FETCH_SIZE x1024 / 8 GiB = {cal_r:.4f} (the gfx950 half-count of wide coalesced reads",
         "described in MI355X_MICROARCH.md §HBM: FETCH_SIZE is doubled below).", "",
         "| kernel | launches | FETCH_SIZE KiB (raw) | read bytes (x2 corrected) | WRITE_SIZE KiB | write bytes |",
         "|---|---|---|---|---|---|"]
out = {}
for k in sorted(set(F) | set(W), key=lambda k: -(sum(F.get(k, [0])) + sum(W.get(k, [0])))):
    f = sum(F.get(k, [0])) / max(1, len(F.get(k, [1])))
    w = sum(W.get(k, [0])) / max(1, len(W.get(k, [1])))
    lines.append(f"| {k[:60]} | {len(F.get(k, []))} | {f:,.0f} | {2 * f * KiB:,.0f} | {w:,.0f} | {w * KiB:,.0f} |")
    out[k] = (2 * f * KiB, w * KiB)
bb = [k for k in out if "k_bb<" in k]
rd = sum(out[k][0] for k in bb) / len(bb)
wr = sum(out[k][1] for k in bb) / len(bb)
pairs = 250 * 4096
alg = pairs * 3 * 8192
lines += ["", f"k_bb per launch: read {rd:,.0f} B + write {wr:,.0f} B = {rd + wr:,.0f} B HBM traffic;",
          f"algorithmic bytes per launch = {pairs} container pairs x 24576 B = {alg:,} B  ->  traffic / algorithmic = {(rd + wr) / alg:.4f}"]
open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump({"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "algorithmic_bytes": alg,
           "pairs_per_launch": pairs, "source": f"profiles/{tag}_pmc_summary.md",
           "note": "FETCH_SIZE doubled per the gfx950 correction; calibrated on k_synth_dir/k_synth_fill"},
          open(os.path.join(ROOT, "profiles", "bb_traffic.json"), "w"), indent=1)
print("\n".join(lines))
