"""Why is k_bb slower inside bench.py than in the microbenchmark harness?  Try: torch or not, pool size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if "--torch" in sys.argv:
    import torch
    torch.cuda.set_device(0)
import croaring_amd
import bench
eng = croaring_amd.Engine(0)
eng.set_timing(True)
for npool in (256, 64):
    pool = eng.pool_synth_bitset(npool, 4096, bench.SEED)
    res = None
    ms = []
    for i in range(12):
        lhs, rhs = bench.schedule(i * 250, 250, npool)
        res = eng.pairwise("and", pool, lhs, pool, rhs, reuse=res)
        st = eng.last_stats()
        ms.append((st["ms_bitset_kernel"], st["ms_total"]))
    k = np.array(ms[2:])
    print(f"torch={'--torch' in sys.argv} pool={npool}: k_bb {k[:,0].mean():.3f} ms ({250*4096*24576/k[:,0].mean()/1e6:.0f} GB/s), call total {k[:,1].mean():.3f} ms", flush=True)
    res.free(); pool.free()
