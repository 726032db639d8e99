"""Where the C2 `or` call spends its time beyond k_bb: host phase clock + kernel list, or vs xor (250 pairs)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import croaring_amd, bench
eng = croaring_amd.Engine(0); eng.set_timing(True)
pool = eng.pool_synth_bitset(256, 4096, bench.SEED)
for op in ("xor", "or", "xor", "or"):
    res = None
    for i in range(6):
        lhs, rhs = bench.schedule(i * 250, 250, 256)
        if i == 2: eng.host_clock(True)
        t0 = time.perf_counter(); res = eng.pairwise(op, pool, lhs, pool, rhs, reuse=res); dt = time.perf_counter() - t0
    st = eng.last_stats()
    print(op, "ms_call", round(dt * 1e3, 3), "ms_total(events)", round(st["ms_total"], 3), "k_bb", round(st["ms_bitset_kernel"], 3),
          "host us/call", [round(x / 4, 1) for x in eng.host_clock()][:6], flush=True)
    del res
