"""Kernel timeline of the LAST many-way call in a rocprofv3 kernel trace (dir given), from its first grouping kernel
(k_many_hist / k_many_gather) to the next one's.  Usage: python scripts/trace_many.py <dir> [title]"""
import csv, glob, os, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_many_hist" in r["Kernel_Name"] or "k_many_gather" in r["Kernel_Name"]]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
print("==", sys.argv[2] if len(sys.argv) > 2 else d)
for r in rows[i0:i1]:
    s = int(r["Start_Timestamp"]) - t0; e = int(r["End_Timestamp"]) - t0
    nm = r["Kernel_Name"].replace("void ", "").split("(")[0]
    print(f"  {nm[:30]:30s} start {s/1e3:8.1f} end {e/1e3:8.1f} dur {(e-s)/1e3:7.1f}  grid {r.get('Grid_Size','?')} wg {r.get('Workgroup_Size','?')}")
print("  call-to-call period", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "us")
