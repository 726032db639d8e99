"""Per-kernel L2 (TCC) hit / miss / memory-request counters of one rocprofv3 --pmc run: argv[1] = output dir, argv[2:] = counters."""
import collections, csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(fs[0])) if fs else []:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
names = sorted({c for v in agg.values() for c in v})
print("| kernel | launches | " + " | ".join(names) + " | hit rate |")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    n = max(cnt[(k, c)] for c in names if (k, c) in cnt)
    hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
    print(f"| `{k[:40]}` | {n} | " + " | ".join(f"{v.get(c, 0.0) / n:.3g}" for c in names) + f" | {100 * hit / max(1.0, hit + miss):.1f} % |")
