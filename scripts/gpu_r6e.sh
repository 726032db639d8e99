cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6e
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -k "many or heap or compat or dropin" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python - <<'P' 2>/dev/null | tee $O/heap_times.txt
import os, sys, time
sys.path.insert(0, "tests")
import numpy as np, torch, croaring_amd
from util import load_bundle, DATASETS
from oracle.pyoracle import Ref
ref = Ref()
eng = croaring_amd.Engine(0)
for name in DATASETS:
    bufs = load_bundle(name)
    pool = eng.pool_from_serialized(bufs)
    hs = [ref.deserialize(b) for b in bufs]
    ts = []
    for _ in range(3):
        t = time.perf_counter(); r = eng.or_many_heap(pool); ts.append(time.perf_counter() - t)
    tc = []
    for _ in range(3):
        t = time.perf_counter(); h = ref.or_many_heap(hs); tc.append(time.perf_counter() - t); 
        same = ref.serialize(h) == r.serialize(0); ref.free(h)
    t = time.perf_counter(); eng.or_many(pool); t_or = time.perf_counter() - t
    print(name, "n", len(bufs), "rhip_or_many_heap ms", round(min(ts) * 1e3, 2), "CRoaring or_many_heap ms", round(min(tc) * 1e3, 2), "bytes equal", same, "rhip_or_many ms", round(t_or * 1e3, 3))
P
echo done
