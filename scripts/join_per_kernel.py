"""Join scripts/per_kernel_c3.py's class statistics with the kernel trace of the same process -> one JSON line per
(op, kernel): items, bytes, stand-alone microseconds, algorithmic GB/s.  argv: <stdout of per_kernel_c3.py> <trace dir>"""
import csv, glob, json, sys
stats = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
rows = sorted(csv.DictReader(open(glob.glob(sys.argv[2] + "/*kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
# the ops run in order: split the trace at the k_class_stats launches (one per op)
segs, cur = [], []
for r in rows:
    cur.append(r)
    if "k_class_stats" in r["Kernel_Name"]:
        segs.append(cur); cur = []
NAMES = {"k_bb": "k_bb<", "k_bba": "k_bba<", "k_ba": "k_ba<", "k_genw": "k_genw", "k_copy": "k_copy", "k_filter": "k_filter",
         "k_wave": "k_wave", "k_probe": "k_probe", "k_usmall": "k_usmall", "k_ivl<32,255>": "k_ivl_all", "k_ivl<8,63>": "k_ivl_all",
         "k_ivl<16,127>": "k_ivl_all"}
for st, seg in zip(stats, segs):
    dur = {}
    for r in seg:
        n = r["Kernel_Name"]
        dur.setdefault(n.split("(")[0].replace("void ", ""), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in st["classes"].items():
        pat = NAMES.get(k, k)
        ds = [x for n, xs in dur.items() if n.startswith(pat) for x in xs]
        if not ds and k in ("k_wave", "k_ba"):  # an X-grouped batch: both classes are served by ONE k_union_g launch
            ds = [x for n, xs in dur.items() if n.startswith("k_union_g") for x in xs]
            k = k + " (in k_union_g)"
        us = sum(ds) / len(ds) / 1e3 if ds else None
        if k.startswith("k_ivl") or k == "k_genw":
            pass  # (k_ivl_all serves three classes in one launch, k_genw is launched twice: durations are per launch)
        print(json.dumps({"dataset": st["dataset"], "op": st["op"], "kernel": k, "items": v["items"], "MB_in": round(v["bytes_in"] / 1e6, 2),
                          "MB_out": round(v["bytes_out"] / 1e6, 2), "us_standalone": round(us, 1) if us else None,
                          "alg_GBps": round((v["bytes_in"] + v["bytes_out"]) / us / 1e3, 1) if us else None}))
