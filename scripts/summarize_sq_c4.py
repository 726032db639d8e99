"""SQ counter pass of the many-way kernels (rocprofv3 --pmc ... on scripts/prof_c4.py) -> markdown table on stdout.
Usage: python scripts/summarize_sq_c4.py <dir of the pass> [title]"""
import collections, csv, glob, os, sys
src = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else "C4 or_many over 100 000 sparse bitmaps"
fs = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
if not fs:
    sys.exit("no counter_collection.csv under " + src)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"# SQ counters of the many-way kernels, {title}\n")
print("`rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS "
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace` on `scripts/prof_c4.py` (6 calls)\n")
print("| kernel | launches | active | wait_any | wait_inst | VALU instr / launch | LDS instr / launch | LDS bank-conflict cycles / LDS-active |")
print("|---|---|---|---|---|---|---|---|")
rows = []
for k, c in agg.items():
    if not k.startswith("k_") or "SQ_WAVE_CYCLES" not in c:
        continue
    n = len(c["SQ_WAVE_CYCLES"]); wc = sum(c["SQ_WAVE_CYCLES"])
    if wc <= 0:
        continue
    f = lambda name: sum(c.get(name, [0]))
    conf = f("SQ_LDS_BANK_CONFLICT") / f("SQ_LDS_IDX_ACTIVE") if f("SQ_LDS_IDX_ACTIVE") else 0.0
    rows.append((wc, f"| `{k}` | {n} | {100 * f('SQ_ACTIVE_INST_ANY') / wc:.0f} % | {100 * f('SQ_WAIT_ANY') / wc:.0f} % | "
                     f"{100 * f('SQ_WAIT_INST_ANY') / wc:.0f} % | {f('SQ_INSTS_VALU') / n:.3g} | {f('SQ_INSTS_LDS') / n:.3g} | {100 * conf:.0f} % |"))
for _, r in sorted(rows, reverse=True):
    print(r)
