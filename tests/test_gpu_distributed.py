"""The sharded or_many / xor_many on real hardware with world > 1: two ranks (two processes, each
with its own engine context) share the one GPU of the test box and talk over a gloo group, so the chunks are
staged through host memory between the REAL rhip_many_partials and rhip_many_finalize; both exchange forms.
The RCCL transport itself is covered with a 1-rank nccl group (the only size one GPU allows) in
test_many_sharded_nccl_world1_dense."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import croaring_amd
        from croaring_amd.distributed import gather_serialized, many_sharded, shard_ids
        from oracle.pyoracle import Oracle
        from util import load_bundle
        oracle = Oracle()
        eng = croaring_amd.Engine(0)
        bufs = load_bundle("census-income")[:60] + load_bundle("wikileaks-noquotes")[:40] + load_bundle("weather_sept_85")[:30]
        hs_all = [oracle.deserialize(b) for b in bufs]
        mine = [int(i) for i in shard_ids(len(bufs), rank, world)]
        pool = eng.pool_from_serialized([bufs[i] for i in mine])
        kmax = max(int(oracle.to_array(h)[-1]) >> 16 for h in hs_all if oracle.cardinality(h))
        ok = []
        for op, fn in (("or", oracle.or_many), ("xor", oracle.xor_many)):
            want = oracle.to_array(fn(hs_all))
            for key_space in (None, kmax + 1):
                owned = many_sharded(eng, pool, op, key_space=key_space)
                hv = oracle.deserialize(owned.serialize(0))
                v = oracle.to_array(hv)
                ok.append(bool(oracle.validate(hv)) and bool(np.all((v >> 16) % world == rank))
                          and bool(np.array_equal(v, want[((want >> 16) % world) == rank])))
                blob = gather_serialized(eng, owned)
                if rank == 0:
                    ok.append(bool(np.array_equal(oracle.to_array(oracle.deserialize(blob)), want)))
        q.put((rank, all(ok), len(ok)))
        dist.barrier()
        eng.close()
    finally:
        dist.destroy_process_group()


def test_many_sharded_two_ranks_one_gpu():
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res


def _worker64(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import croaring_amd
        from croaring_amd.distributed import gather_serialized, many_sharded, shard_ids
        from oracle.pyoracle import Oracle
        from util import GOLD, c5_inputs
        oracle = Oracle()
        eng = croaring_amd.Engine(0)
        gold = np.load(os.path.join(GOLD, "c5_wikileaks64_pairs.npz"))
        bufs = c5_inputs()
        mine = [int(i) for i in shard_ids(len(bufs), rank, world)]
        pool = eng.pool_from_serialized64([bufs[i] for i in mine])
        owned = many_sharded(eng, pool, "or")  # 48-bit keys: sparse exchange
        vals, _ = owned.to_values()
        ok = [owned.is64, bool(np.all(((vals >> np.uint64(16)) % np.uint64(world)) == np.uint64(rank)))]
        card = torch.tensor([int(owned.cardinalities()[0])], dtype=torch.int64)
        dist.all_reduce(card)
        ok.append(int(card.item()) == int(gold["fold_or_card"][0]))
        blob = gather_serialized(eng, owned)
        if rank == 0:
            hg = oracle.deserialize64(blob)
            hw = oracle.deserialize64(bytes(gold["fold_or"]))
            x = oracle.op64("xor", hg, hw)
            ok.append(oracle.cardinality64(x) == 0)
        q.put((rank, all(ok), len(ok)))
        dist.barrier()
        eng.close()
    finally:
        dist.destroy_process_group()


def test_many_sharded64_two_ranks_one_gpu():
    """BASELINE configs[4]: the C5 200-way roaring64 union sharded b mod 2 over two ranks (two processes on the one
    GPU, gloo, chunks staged through host), against the reference's fold fixture (tests/golden/c5_wikileaks64_pairs.npz)."""
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker64, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res


def test_many_sharded_nccl_world1_dense(engine, oracle):
    """The dense (fixed-shape all_to_all_single) and sparse exchange on the RCCL backend with a 1-rank group,
    issued on the engine's stream with no host synchronisation between the stages."""
    import torch
    import torch.distributed as dist
    from croaring_amd.distributed import gather_serialized, many_sharded
    from util import load_bundle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        bufs = load_bundle("weather_sept_85")[:64]
        hs = [oracle.deserialize(b) for b in bufs]
        pool = engine.pool_from_serialized(bufs)
        for op, fn in (("or", oracle.or_many), ("xor", oracle.xor_many)):
            want = fn(hs)
            for key_space in (None, 64, 4096):
                for _ in range(3):  # repeated calls recycle the partial-chunk buffers through the context cache
                    owned = many_sharded(engine, pool, op, key_space=key_space)
                hg = oracle.deserialize(gather_serialized(engine, owned))
                assert oracle.validate(hg)
                assert np.array_equal(oracle.to_array(hg), oracle.to_array(want)), (op, key_space)
                oracle.free(hg)
            oracle.free(want)
        for h in hs:
            oracle.free(h)
    finally:
        if created:
            dist.destroy_process_group()


def _rccl():
    """librccl through ctypes: the test owns the communicator, as a C / Go caller of rhip_many_sharded would."""
    import ctypes as C
    import torch  # noqa  (its bundled librccl is then already in the process; dlopen by name finds it)
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            return C.CDLL(name, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    pytest.skip("no librccl on this box")


@pytest.mark.parametrize("key_space", [4096, 0])
def test_many_sharded_c_abi_world1(engine, oracle, key_space):
    """rhip_many_sharded (the C entry point: stage 1 -> RCCL exchange -> stage 3 inside the library) on a one-rank
    communicator -- the only size one GPU allows: dense (with the all-to-all really issued) and sparse (all-gathers,
    packing, grouped send / recv degenerate to the local copy), bytes equal to rhip_or_many / the oracle's or_many."""
    import ctypes as C
    import croaring_amd
    from gen_inputs import random_bitmap
    L = _rccl()

    class UID(C.Structure):
        _fields_ = [("b", C.c_char * 128)]
    uid = UID()
    L.ncclGetUniqueId.argtypes = [C.POINTER(UID)]
    assert L.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
    assert L.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        rng = np.random.default_rng(11)
        vals = [random_bitmap(rng) for _ in range(40)]
        vals = [v[v < (4096 << 16)] for v in vals]  # keys below the dense key space
        hs = [oracle.from_sorted(v) for v in vals]
        bufs = [oracle.serialize(h) for h in hs]
        import os
        os.environ["RHIP_SHARD_FORCE_COLLECTIVE"] = "1"
        eng = croaring_amd.Engine(0)
        os.environ.pop("RHIP_SHARD_FORCE_COLLECTIVE")
        try:
            pool = eng.pool_from_serialized(bufs)
            for op, fn in (("or", oracle.or_many), ("xor", oracle.xor_many)):
                want = fn(hs)
                got = eng.many_sharded_native(comm.value, op, pool, None, key_space)
                hv = oracle.deserialize(got.serialize(0))
                assert oracle.validate(hv)
                assert np.array_equal(oracle.to_array(hv), oracle.to_array(want)), (op, key_space)
                sub = [3, 17, 5, 5, 29]
                ws = fn([hs[i] for i in sub])
                gs = eng.many_sharded_native(comm.value, op, pool, sub, key_space)
                assert np.array_equal(oracle.to_array(oracle.deserialize(gs.serialize(0))), oracle.to_array(ws))
        finally:
            eng.close()
    finally:
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclCommDestroy(comm)


@pytest.mark.parametrize("world,key_space,bits", [(2, 4096, 32), (4, 0, 32), (8, 4096, 32), (8, 0, 32), (3, 0, 64), (8, 0, 64)])
def test_many_sharded_c_abi_multi_rank(world, key_space, bits):
    """rhip_many_sharded at world sizes 2 .. 8 on ONE GPU: the ranks are threads of a child process, each with its own
    context and its share of the bitmaps (b mod world), and librccl is tests/fake_rccl/libfake_rccl.so (RHIP_RCCL_LIB) -- a
    test-only stand-in whose collectives are barriers + device copies (the real RCCL refuses two ranks on one device).  What
    this executes, and nothing else did before the 8-GPU scaling run: the owner partition, the packing and the send /
    receive offsets of the SPARSE exchange at world > 1, the all-to-all shape of the dense one, the owners' key sets
    (every owner holds exactly the keys = rank mod world) and their union against the reference's or_many / xor_many /
    the 64-bit fold."""
    import shutil
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl")
    lib = os.path.join(here, "libfake_rccl.so")
    src = os.path.join(here, "fake_rccl.cpp")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.run([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "--offload-arch=gfx950", src, "-o", lib], check=True)
    env = dict(os.environ, RHIP_RCCL_LIB=lib)
    p = subprocess.run([sys.executable, os.path.join(here, "child.py"), str(world), str(key_space), str(bits)], capture_output=True,
                       text=True, timeout=300, env=env)
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-3000:])
