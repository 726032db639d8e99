"""SQ counter passes of scripts/gpu_final_r2.sh (rocprofv3 --pmc ... --kernel-trace on scripts/prof_weather.py) ->
profiles/<tag>_pmc_weather_sq.md.  Usage: python scripts/summarize_sq.py <dir with pmc_w_and / pmc_w_or> <tag>"""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
out = [f"# SQ counters of the class kernels, C3 weather_sept_85 all-pairs ({tag})", "",
       "`rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS "
       "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace` on `scripts/prof_weather.py {and,or}` (12 batches each).",
       "Shares are of SQ_WAVE_CYCLES; `active` = an instruction of the wave is issuing, `wait_any` = parked on s_waitcnt / "
       "barrier, `wait_inst` = issue stall.  VALU / LDS = wave-level instructions per launch.", ""]
for op in ("and", "or"):
    fs = glob.glob(os.path.join(src, f"pmc_w_{op}", "*counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out += [f"## {op}", "", "| kernel | launches | active | wait_any | wait_inst | VALU instr / launch | LDS instr / launch | "
            "LDS bank-conflict cycles / LDS-active |", "|---|---|---|---|---|---|---|---|"]
    rows = []
    for k, c in agg.items():
        if not k.startswith("k_") or "SQ_WAVE_CYCLES" not in c:
            continue
        n = len(c["SQ_WAVE_CYCLES"])
        wc = sum(c["SQ_WAVE_CYCLES"])
        if wc <= 0:
            continue
        f = lambda name: sum(c.get(name, [0]))
        conf = f("SQ_LDS_BANK_CONFLICT") / f("SQ_LDS_IDX_ACTIVE") if f("SQ_LDS_IDX_ACTIVE") else 0.0
        rows.append((wc, f"| `{k}` | {n} | {100 * f('SQ_ACTIVE_INST_ANY') / wc:.0f} % | {100 * f('SQ_WAIT_ANY') / wc:.0f} % | "
                         f"{100 * f('SQ_WAIT_INST_ANY') / wc:.0f} % | {f('SQ_INSTS_VALU') / n:.3g} | {f('SQ_INSTS_LDS') / n:.3g} | {100 * conf:.0f} % |"))
    out += [r for _, r in sorted(rows, reverse=True)] + [""]
open(os.path.join(ROOT, "profiles", f"{tag}_pmc_weather_sq.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
