"""Copy what profiles/ cites out of gpurun_out/final_rN (scratch) after `gpurun -- 'bash scripts/gpu_final_rN.sh'`
(argv[1] = tag, default r03):
kernel-stats CSVs, the bench line, JSON-lines tables, the kernel timelines of one batch per workload, and the PMC
summaries (scripts/summarize_pmc.py, scripts/summarize_sq.py)."""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC = os.path.join(ROOT, "gpurun_out", "final_" + TAG.replace("0", "", 1))
DST = os.path.join(ROOT, "profiles")

def cp(src, dst):
    if os.path.exists(src):
        if os.path.getsize(src) == 0:  # (round 3 committed an empty r03_c2_ops.jsonl that DESIGN cited: never again)
            raise SystemExit(f"collect_profiles: {src} is EMPTY -- the step that writes it failed; fix it before citing it")
        shutil.copy(src, os.path.join(DST, dst))
        print("  ", dst)

cp(os.path.join(SRC, "bench.json"), f"{TAG}_bench.json")
cp(os.path.join(SRC, "bench_detail.json"), f"{TAG}_bench_detail.json")
cp(os.path.join(SRC, "quick_all.txt"), f"{TAG}_quick_all.txt")
cp(os.path.join(SRC, "pmc_l2.md"), f"{TAG}_pmc_l2.md")
for f, d in (("c2_ops.jsonl", "c2_ops"), ("quick_c3.jsonl", "quick_c3"), ("class_throughput.jsonl", "class_throughput"),
             ("poolops.jsonl", "poolops"), ("per_kernel_c3.jsonl", "per_kernel_c3"), ("multi.txt", "multi_ops")):
    cp(os.path.join(SRC, f), f"{TAG}_{d}" + (".txt" if f.endswith(".txt") else ".jsonl"))
for f in ("pmc_c4_sq.md", "placement_runs.txt", "timelines_many.txt"):
    cp(os.path.join(SRC, f), f"{TAG}_{f}")
for d in sorted(glob.glob(os.path.join(SRC, "prof_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[5:]
    st = glob.glob(os.path.join(d, "*kernel_stats.csv"))
    if st:
        out = {"bench": f"{TAG}_bench_c2_kernel_stats.csv"}.get(name, f"{TAG}_{name.replace('w_', 'c3_')}_kernel_stats.csv")
        cp(st[0], out)
# one batch per workload as a timeline (start / end / duration of every kernel, microseconds from the batch's k_count)
lines = [f"# Kernel timelines of ONE batched call per workload ({TAG}; rocprofv3 --kernel-trace, scripts/prof_weather.py)",
         "# start / end / duration in microseconds from the first kernel of the call; all-pairs batches", ""]
for d in sorted(glob.glob(os.path.join(SRC, "prof_*"))):
    tr = glob.glob(os.path.join(d, "*kernel_trace.csv"))
    if not tr:
        continue
    rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_count" in r["Kernel_Name"] or "k_plan_restore" in r["Kernel_Name"] or "k_many_gather" in r["Kernel_Name"]]  # (k_plan_restore: a batch whose plan came from its pair list, round 6)
    if len(idx) < 2:
        continue
    i0, i1 = idx[-2], idx[-1]
    t0 = int(rows[i0]["Start_Timestamp"])
    log = os.path.join(SRC, os.path.basename(d) + ".log")
    ms = [l for l in open(log) if "min ms" in l] if os.path.exists(log) else []
    lines.append(f"== {os.path.basename(d)[5:]}   ({ms[0].strip() if ms else ''})")
    for r in rows[i0:i1]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        lines.append(f"  {r['Kernel_Name'].split('(')[0].replace('void ', '')[:28]:28s} start {s / 1e3:8.1f}  end {e / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}")
    lines.append(f"  call-to-call period {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us")
    lines.append("")
open(os.path.join(DST, f"{TAG}_timelines.txt"), "w").write("\n".join(lines))
print("  ", f"{TAG}_timelines.txt")
if os.path.isdir(os.path.join(SRC, "pmc_fetch")):
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "summarize_pmc.py"), SRC, TAG], check=False, stdout=subprocess.DEVNULL)
    print("  ", f"{TAG}_pmc_summary.md, bb_traffic.json")
if os.path.isdir(os.path.join(SRC, "pmc_w_and")):
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "summarize_sq.py"), SRC, TAG], check=False, stdout=subprocess.DEVNULL)
    print("  ", f"{TAG}_pmc_weather_sq.md")
