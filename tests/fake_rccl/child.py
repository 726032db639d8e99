"""Child of tests/test_gpu_distributed.py::test_many_sharded_c_abi_multi_rank: rhip_many_sharded at world = argv[1] on ONE GPU,
the ranks being threads of this process over tests/fake_rccl/libfake_rccl.so (RHIP_RCCL_LIB, set by the parent).
argv[2] = key space (0: the sparse exchange), argv[3] = 32 | 64.  Prints OK or raises."""
import ctypes as C
import os
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
world, key_space, bits = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
import torch  # noqa: F401
import croaring_amd
from oracle.pyoracle import Oracle
from gen_inputs import random_bitmap
from util import c5_inputs, load_bundle

F = C.CDLL(os.environ["RHIP_RCCL_LIB"], mode=C.RTLD_GLOBAL)
F.fake_group_create.restype = C.c_void_p
F.fake_group_create.argtypes = [C.c_int]
F.fake_comm_create.restype = C.c_void_p
F.fake_comm_create.argtypes = [C.c_void_p, C.c_int]
F.fake_set_self.argtypes = [C.c_void_p]
oracle = Oracle()
if bits == 64:
    bufs = c5_inputs(load_bundle("wikileaks-noquotes")[:48])
else:
    rng = np.random.default_rng(5)
    vals = [random_bitmap(rng) for _ in range(48)]
    if key_space:
        vals = [v[v < (key_space << 16)] for v in vals]
    bufs = [oracle.serialize(oracle.from_sorted(v)) for v in vals]
group = F.fake_group_create(world)
out, errs = {}, []


def rank_main(rank):
    try:
        eng = croaring_amd.Engine(0)
        mine = [b for i, b in enumerate(bufs) if i % world == rank]
        pool = eng.pool_from_serialized64(mine) if bits == 64 else eng.pool_from_serialized(mine)
        comm = F.fake_comm_create(group, rank)
        F.fake_set_self(comm)
        for op in ("or", "xor"):
            got = eng.many_sharded_native(comm, op, pool, None, key_space)
            out[(op, rank)] = got.serialize(0)
            if bits == 64:
                v, _ = got.to_values()
                assert np.all(((v >> np.uint64(16)) % np.uint64(world)) == np.uint64(rank)), (op, rank, "holds a key it does not own")
        eng.synchronize()
    except Exception as e:  # (a rank that dies leaves the others at a barrier: the parent's timeout ends the run)
        errs.append((rank, repr(e)))
        raise


ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in ts:
    t.start()
for t in ts:
    t.join()
assert not errs, errs
eng = croaring_amd.Engine(0)
for op in ("or", "xor"):
    owned = [out[(op, r)] for r in range(world)]
    if bits == 64:
        hs = [oracle.deserialize64(b) for b in bufs]
        acc = oracle.deserialize64(bufs[0])
        for h in hs[1:]:
            nxt = oracle.op64(op, acc, h)
            oracle.free64(acc)
            acc = nxt
        got = None
        for r, b in enumerate(owned):
            hb = oracle.deserialize64(b)
            nxt = hb if got is None else oracle.op64("or", got, hb)
            if got is not None:
                oracle.free64(got); oracle.free64(hb)
            got = nxt
        x = oracle.op64("xor", got, acc)
        assert oracle.cardinality64(x) == 0, (op, "union of the owners' results differs from the fold")
    else:
        hs = [oracle.deserialize(b) for b in bufs]
        want = (oracle.or_many if op == "or" else oracle.xor_many)(hs)
        parts = []
        for r, b in enumerate(owned):
            hb = oracle.deserialize(b)
            assert oracle.validate(hb)
            v = oracle.to_array(hb)
            assert np.all(((v >> 16) % world) == r), (op, r, "holds a key it does not own")
            parts.append(v)
        got = np.sort(np.concatenate(parts))
        assert np.array_equal(got, oracle.to_array(want)), (op, "union of the owners' results differs from the reference's")
print("OK", world, key_space, bits)
