"""Per-class kernel throughput on synthetic single-type pools: array x array, array x bitset, bitset x array,
run x run ... (JSON lines).  python scripts/quick_classes.py
PAIRS="R100 x R100,R100 x A874" restricts the pairs; STATS=1 adds the per-class item / byte counts of every batch."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd

eng = croaring_amd.Engine(0)
rng = np.random.default_rng(3)
NB, NC = 48, 64


def pool(kind, card):
    lists = []
    for _ in range(NB):
        parts = []
        for k in range(NC):
            if kind == "array":
                v = np.sort(rng.choice(65536, int(card * rng.uniform(0.5, 1.5)), replace=False))
            elif kind == "bitset":
                v = np.flatnonzero(rng.random(65536) < 0.5)
            else:  # runs
                cuts = np.sort(rng.choice(65536, 2 * card, replace=False))
                v = np.concatenate([np.arange(cuts[2 * i], cuts[2 * i + 1]) for i in range(card)])
            parts.append((np.uint32(k) << np.uint32(16)) | v.astype(np.uint32))
        lists.append(np.concatenate(parts))
    p = eng.pool_from_values(lists)
    return eng.run_optimize(p) if kind == "runs" else p


pools = {"A874": pool("array", 874), "A200": pool("array", 200), "A3000": pool("array", 3000), "B": pool("bitset", 0),
         "R100": pool("runs", 100)}
for k, p in pools.items():
    print(json.dumps({"pool": k, "types(B,A,R)": p.type_counts()}), flush=True)
lhs, rhs = np.meshgrid(np.arange(NB, dtype=np.uint32), np.arange(NB, dtype=np.uint32))
lhs, rhs = lhs.ravel().copy(), rhs.ravel().copy()
items = lhs.size * NC
only = [x.strip() for x in os.environ.get("PAIRS", "").split(",") if x.strip()]
for a, b in (("A874", "A874"), ("A200", "A200"), ("A3000", "A3000"), ("A874", "B"), ("B", "A874"), ("A200", "B"),
             ("R100", "R100"), ("R100", "A874"), ("R100", "B"), ("B", "B")):
    if only and f"{a} x {b}" not in only:
        continue
    row = {"pair": f"{a} x {b}", "items": items}
    for op in ("and", "or", "xor", "andnot"):
        res, ts = None, []
        for _ in range(5):
            t = time.perf_counter()
            res = eng.pairwise(op, pools[a], lhs, pools[b], rhs, reuse=res)
            ts.append(time.perf_counter() - t)
        st = eng.last_stats()
        if os.environ.get("STATS"):
            eng.set_class_stats(True)
            eng.pairwise(op, pools[a], lhs, pools[b], rhs)
            eng.set_class_stats(False)
            row.setdefault("classes", {})[op] = {k: v["items"] for k, v in eng.last_class_stats().items() if v["items"]}
        tm = min(ts[1:])
        row[op] = {"ms": round(tm * 1e3, 3), "ns_per_item": round(tm / items * 1e9, 2),
                   "TBps": round((st["bytes_in"] + st["bytes_out"]) / tm / 1e12, 3)}
    print(json.dumps(row), flush=True)
