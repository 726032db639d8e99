"""Kernel-logic tests WITHOUT a GPU: the engine sources of croaring_amd/csrc compiled against the hipemu shim
(tests/emu: a wave64 SIMT emulator -- test infrastructure, never part of the product) run the same parity
bodies as tests/test_gpu_parity.py, against the same golden fixtures and the same oracle.

What this can and cannot show: it executes the real kernel source (indexing, LDS protocols, ballots /
shuffles, queue planning, result typing, serialization) and holds it to a stricter cross-lane ordering
model than the hardware (lanes of a wave are only ordered by collectives and barriers).  It says nothing
about performance, memory coalescing or hardware-only behaviour; `-m gpu` on an MI355X stays the parity
gate."""
import numpy as np
import pytest

import test_gpu_parity as G
import test_gpu_64bit as G64
from util import OPS, crc, load_bundle, load_pairs

synth = G.synth  # module-scoped fixture, shared with the GPU suite


@pytest.fixture(scope="module")
def emu():
    from emu import build_emu, emu_engine
    if not __import__("os").path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++ to compile the kernels for the host")
    eng = emu_engine()
    yield eng
    eng.close()


@pytest.mark.parametrize("op", OPS)
def test_emu_synth_every_type_pair(emu, oracle, synth, op):
    G.test_synth_every_type_pair(emu, oracle, synth, op)


@pytest.mark.parametrize("name", ["census1881", "weather_sept_85", "wikileaks-noquotes"])
def test_emu_realdata_pair_sample(emu, name):
    """A seeded sample of the all-pairs fixture (the full 19 900-pair sets are the GPU suite's job):
    cardinality, portable size and crc32 of every sampled result equal the reference's."""
    bufs = load_bundle(name)
    gold = load_pairs(name)
    pool = emu.pool_from_serialized(bufs)
    rng = np.random.default_rng(5)
    sel = np.sort(rng.choice(len(gold["pairs"]), 700, replace=False))
    lhs = gold["pairs"][sel, 0].astype(np.uint32)
    rhs = gold["pairs"][sel, 1].astype(np.uint32)
    for op in OPS:
        res = emu.pairwise(op, pool, lhs, pool, rhs)
        cards = res.cardinalities()
        assert np.array_equal(cards, gold[f"{op}_card"][sel].astype(np.uint64)), (name, op)
        assert np.array_equal(emu.pairwise_cardinality(op, pool, lhs, pool, rhs), cards), (name, op)
        bad = [k for k in range(len(sel)) if len(s := res.serialize(k)) != gold[f"{op}_size"][sel[k]]
               or crc(s) != gold[f"{op}_crc"][sel[k]]]
        assert not bad, f"{name} {op}: {len(bad)} of {len(sel)} sampled results differ from the reference"


def test_emu_lane_order_independence(emu, oracle, synth):
    """Same batch with the emulator resuming the lanes of each wave in shuffled order: any cross-lane LDS
    exchange that is not ordered by a collective or barrier changes the result."""
    try:
        for seed in (1, 2):
            emu.lib.hipemu_set_shuffle(seed)
            for op in OPS:
                G.test_synth_every_type_pair(emu, oracle, synth, op)
            G.test_synth_many(emu, oracle, synth)
    finally:
        emu.lib.hipemu_set_shuffle(0)


@pytest.mark.parametrize("name", ["census1881", "weather_sept_85"])
def test_emu_realdata_many(emu, oracle, name):
    G.test_realdata_many(emu, oracle, name)


def test_emu_synth_many(emu, oracle, synth):
    G.test_synth_many(emu, oracle, synth)


def test_emu_bitset_only_synthetic_pool(emu, oracle):
    G.test_bitset_only_synthetic_pool(emu, oracle)


def test_emu_edge_cases(emu, oracle):
    G.test_edge_cases(emu, oracle)


def test_emu_or_many_full_container_typing(emu, oracle):
    G.test_or_many_full_container_typing(emu, oracle)


def test_emu_directory_and_payload_extremes(emu, oracle):
    G.test_directory_and_payload_extremes(emu, oracle)


def test_emu_array_array_union_boundaries(emu, oracle):
    G.test_array_array_union_boundaries(emu, oracle)


def test_emu_randomized_pools(emu, oracle):
    G.test_randomized_pools(emu, oracle, 11)


def test_emu_64bit(emu, oracle):
    G64.test_64bit_pairwise_and_many(emu, oracle)
    G64.test_64bit_reference_fixtures(emu)


def test_emu_dropin_entry_points(emu, ref):
    """The CRoaring-named per-call drop-ins (include/roaring_hip_compat.h) on reference-made structs."""
    import test_gpu_compat as GC
    lib = GC.bind(emu.lib)
    GC.test_dropin_pairwise(lib, ref)
    GC.test_dropin_many(lib, ref)
    GC.test_dropin_lazy_family(lib, ref)


def test_emu_pinned_allocator_hook(emu, ref):
    """tests/hook_child.py against the emulator build (the arena is plain host memory there: the allocator, the hook
    plumbing and the drop-ins' use of the reference's roaring_malloc / roaring_free are what is tested)."""
    import os, subprocess, sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hook_child.py")
    p = subprocess.run([sys.executable, child, "emu"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "hook ok" in p.stdout, (p.stdout + p.stderr)[-3000:]


def test_emu_pool_reshaping(emu, oracle):
    """rhip_pool_select / rhip_pairwise_inplace / run_optimize / remove_run_compression / predicates."""
    import test_gpu_poolops as GP
    GP.test_pool_select(emu, oracle)
    GP.test_pool_select_64bit(emu, oracle)
    for op in OPS:
        GP.test_pairwise_inplace(emu, oracle, op)
    GP.test_pairwise_inplace_chain(emu, oracle)
    for mode in ("run_optimize", "remove_run_compression"):
        GP.test_container_conversions(emu, oracle, mode)
    GP.test_pairwise_predicates(emu, oracle)
    GP.test_bulk_serialization(emu, oracle)


def test_emu_device_deserialization(emu, oracle):
    import test_gpu_poolops as GP
    GP.test_device_deserialization_roundtrip(emu, oracle)
    GP.test_device_deserialization_64bit(emu, oracle)
    GP.test_device_deserialization_rejects_what_the_host_loader_rejects(emu, oracle)
    GP.test_device_deserialization_fuzz_64bit(emu, oracle)


def test_emu_payload_layout_16_vs_lines(oracle, monkeypatch):
    from emu import build_emu, emu_engine
    if not __import__("os").path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++ to compile the kernels for the host")
    import test_gpu_poolops as GP
    GP.payload_layout_body(emu_engine, oracle, monkeypatch)


def test_emu_frozen_format(emu, oracle):
    import test_gpu_poolops as GP
    GP.frozen_body(emu, oracle)


def test_emu_flip(emu, oracle):
    import test_gpu_poolops as GP
    GP.test_flip(emu, oracle)
    GP.test_flip_64bit(emu, oracle)


def test_emu_value_lists(emu, oracle):
    import test_gpu_poolops as GP
    GP.test_value_lists_roundtrip(emu, oracle)
    GP.test_value_lists_64bit(emu, oracle)


def test_emu_pool_reshaping_vs_reference(emu, ref):
    import test_gpu_poolops as GP
    GP.test_container_conversions_vs_reference(emu, ref)
    GP.test_pairwise_predicates_vs_reference(emu, ref)


@pytest.mark.parametrize("op", ["or", "xor"])
def test_emu_sharded_many_partials_and_finalize(emu, oracle, op):
    """The two kernels of the multi-GPU or_many / xor_many with G logical shards: per-shard partial
    chunks (rhip_many_partials) -> chunks routed to key owners -> owner combine + canonicalise (rhip_many_finalize).
    Under the emulator device memory is host memory, so the routing that torch does on the GPU is numpy here."""
    import ctypes as C
    from croaring_amd.distributed import shard_ids
    from util import load_bundle
    bufs = load_bundle("census-income")[:48] + load_bundle("wikileaks-noquotes")[:24]
    hs = [oracle.deserialize(b) for b in bufs]
    pool = emu.pool_from_serialized(bufs)
    want = (oracle.or_many if op == "or" else oracle.xor_many)(hs)
    for shards in (2, 5):
        parts = [emu.many_partials(op, pool, shard_ids(len(bufs), s, shards)) for s in range(shards)]
        emu.synchronize()
        keys, words = [], []
        for p in parts:
            if p.n_keys:
                keys.append(np.ctypeslib.as_array(C.cast(p.d_keys, C.POINTER(C.c_uint64)), (p.n_keys,)).copy())
                words.append(np.ctypeslib.as_array(C.cast(p.d_words, C.POINTER(C.c_uint64)), (p.n_keys, 1024)).copy())
        K, W = np.concatenate(keys), np.concatenate(words)
        got = []
        for owner in range(shards):
            sel = (K % np.uint64(shards)) == owner
            k, w = np.ascontiguousarray(K[sel]), np.ascontiguousarray(W[sel])
            res = emu.many_finalize(op, False, k.size, k.ctypes.data if k.size else 0, w.ctypes.data if k.size else 0)
            h = oracle.deserialize(res.serialize(0))
            assert oracle.validate(h)
            v = oracle.to_array(h)
            assert np.all((v >> 16) % shards == owner)
            got.append(v)
            oracle.free(h)
        assert np.array_equal(np.sort(np.concatenate(got)), oracle.to_array(want)), (op, shards)
        for p in parts:
            p.free()
    for h in hs + [want]:
        oracle.free(h)


def test_emu_pairwise_multi(emu, oracle, synth):
    G.test_pairwise_multi(emu, oracle, synth)


@pytest.mark.parametrize("mode", ["1"])
def test_emu_pairwise_multi_explicit_units(oracle, synth, monkeypatch, mode):
    from emu import build_emu, emu_engine
    if not __import__("os").path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++ to compile the kernels for the host")
    monkeypatch.setenv("RHIP_EXPLICIT_UNITS", mode)
    eng = emu_engine()
    try:
        G.test_pairwise_multi(eng, oracle, synth)
    finally:
        eng.close()


def test_emu_class_stats(emu, oracle, synth):
    G.test_class_stats(emu, oracle, synth)


def test_emu_sparse_many_and_dense_shards(emu, oracle):
    G.sparse_many_body(emu, oracle, n=400, worlds=(1, 3))


@pytest.mark.parametrize("ch", ["1024", "3", "rev"])
def test_emu_many_unit_shapes(oracle, synth, monkeypatch, ch):
    """RHIP_MANY_CH forces the piece size of the many-way path: 1024 = pieces of more than one 512-member staging
    chunk (key 0 of the sparse set has one member per bitmap), 3 = nearly every group cut into partial chunks;
    "rev": the counting sort's scatter fills its reservations backwards in workgroups of 1 024 members -- the order
    inside a group is arbitrary on the GPU (atomics), and the emulator's is otherwise always the gathered order: the
    full-union typing must come from the members' tags."""
    from emu import build_emu, emu_engine
    if not __import__("os").path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++ to compile the kernels for the host")
    if ch == "rev":
        monkeypatch.setenv("RHIP_MANY_REVERSE", "1")
        monkeypatch.setenv("RHIP_MANY_T", "1024")
        monkeypatch.setenv("RHIP_MANY_CH", "5")
    else:
        monkeypatch.setenv("RHIP_MANY_CH", ch)
    eng = emu_engine()
    try:
        G.sparse_many_body(eng, oracle, n=700 if ch == "1024" else 150, worlds=(2,))
        G.test_synth_many(eng, oracle, synth)
        G.test_or_many_full_container_typing(eng, oracle)
    finally:
        eng.close()


@pytest.mark.parametrize("pf", ["2", "4"])
def test_emu_many_ring_depths(oracle, synth, monkeypatch, pf):
    """RHIP_MANY_PF: k_many_l1's ring of load buffers at other depths than the default (1) -- PF + 1 buffers unrolled over
    one turn, the tail of every octet's range running past its end."""
    from emu import build_emu, emu_engine
    if not __import__("os").path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++ to compile the kernels for the host")
    monkeypatch.setenv("RHIP_MANY_PF", pf)
    eng = emu_engine()
    try:
        G.sparse_many_body(eng, oracle, n=200, worlds=(2,))
        G.test_synth_many(eng, oracle, synth)
    finally:
        eng.close()


def test_emu_array_filter_probe_boundaries(emu, oracle):
    G.test_array_filter_probe_boundaries(emu, oracle)


def test_emu_array_array_union_boundaries(emu, oracle):
    G.test_array_array_union_boundaries(emu, oracle)


def test_emu_tiny_interval_pairs(emu, oracle):
    G.test_tiny_interval_pairs(emu, oracle)


def test_emu_prepared_pair_lists(emu, oracle, synth):
    G.test_prepared_pair_lists(emu, oracle, synth)


def test_emu_tiny_passthrough_containers(emu, oracle):
    G.test_tiny_passthrough_containers(emu, oracle)


def test_emu_long_interval_lists(emu, oracle):
    G.test_long_interval_lists(emu, oracle)


def test_emu_pairwise_placed(emu, oracle, synth):
    G.test_pairwise_placed(emu, oracle, synth)


def test_emu_many_long_run_passthrough(emu, oracle):
    G.test_many_long_run_passthrough(emu, oracle)


def test_emu_xor_many_fold_typing(emu, oracle):
    G.xor_many_typing_body(emu, oracle, iters=36)


def test_emu_many_selection_with_a_wide_bitmap(emu, oracle):
    G.test_many_selection_with_a_wide_bitmap(emu, oracle)


def test_emu_robust_deserialization_corpus(emu):
    import test_gpu_poolops as GP
    from oracle.pyoracle import Ref
    GP.robust_corpus_body(emu, Ref() if Ref.available() else None)


def test_emu_inplace_dropins_survive_transient_alloc_failure(emu, ref):
    import os, sys
    import test_gpu_compat as GC
    lib = GC.bind(emu.lib)
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hook_child.py")
    GC.alloc_failure_body(lib, ref, [sys.executable, child, "emu", "allocfail"])


def test_emu_64bit_many_grouping_paths(oracle, monkeypatch):
    from emu import build_emu, emu_engine
    if not __import__("os").path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++ to compile the kernels for the host")
    G64.many64_grouping_body(emu_engine, oracle, monkeypatch)


def test_emu_or_many_heap_tournament(emu, oracle):
    G.or_many_heap_body(emu, oracle, iters=30)


def test_emu_join_fallback(oracle, synth, monkeypatch):
    """A forked batch whose flag gate reports a time-out (RHIP_JOIN_FAIL=1) is finished through the fallback of
    rhip_pairwise_end -- streams waited for, the tail's scratch cleared, the tail run again -- with the same bytes."""
    from emu import build_emu, emu_engine
    if not __import__("os").path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++ to compile the kernels for the host")
    monkeypatch.setenv("RHIP_FORK_MIN_MB", "0")
    monkeypatch.setenv("RHIP_SPIN_JOIN", "2")
    monkeypatch.setenv("RHIP_JOIN_FAIL", "1")
    eng = emu_engine()
    try:
        G.test_synth_every_type_pair(eng, oracle, synth, "or")
        assert eng.join_recovered() >= 1
        G.test_edge_cases(eng, oracle)
        G.test_batches_in_flight(eng, oracle, synth)
    finally:
        eng.close()


@pytest.mark.parametrize("mode", ["1", "2", "fork", "nomerge"])
def test_emu_explicit_unit_arrays(oracle, synth, monkeypatch, mode):
    """Batches whose bitmaps all have <= 256 containers plan on implicit units (unit = pair / 2 pair + side), four
    units per wave up to 64 containers; RHIP_EXPLICIT_UNITS=1 forces the staged unit arrays that larger bitmaps need,
    =2 implicit units one per wave, on the same small inputs."""
    from emu import build_emu, emu_engine
    if not __import__("os").path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++ to compile the kernels for the host")
    if mode == "fork":       # the stand-alone class kernels (a small batch otherwise runs them as ONE launch, k_classes)
        monkeypatch.setenv("RHIP_FORK_MIN_MB", "0")
        monkeypatch.setenv("RHIP_SPIN_JOIN", "2")  # (the emulator runs kernels one by one: no self-test, the flag join forced on)
    elif mode == "nomerge":
        monkeypatch.setenv("RHIP_MERGE_CLASSES", "0")
    else:
        monkeypatch.setenv("RHIP_EXPLICIT_UNITS", mode)
    eng = emu_engine()
    try:
        G.test_edge_cases(eng, oracle)
        G.test_synth_every_type_pair(eng, oracle, synth, "xor")
        G.test_synth_every_type_pair(eng, oracle, synth, "andnot")
        if mode == "nomerge":
            G.test_tiny_passthrough_containers(eng, oracle)
        if mode == "1":
            G.test_prepared_pair_lists(eng, oracle, synth)
        if mode in ("fork", "nomerge"):
            G.test_synth_every_type_pair(eng, oracle, synth, "and")
            G.test_synth_every_type_pair(eng, oracle, synth, "or")
            G.test_pairwise_multi(eng, oracle, synth)
    finally:
        eng.close()


@pytest.mark.parametrize("vmm", ["1", "0"])
def test_emu_arena_placement_forced(oracle, synth, monkeypatch, vmm):
    """place_arena on small arenas (RHIP_ARENA_PLACE_MIN_MB=0) through the emulator: the bookkeeping -- by address (the
    shim's hipMemCreate / hipMemMap are a memfd mapped at the probed positions: the same bytes wherever they are mapped)
    and by candidates (allocate, probe, keep one, release the rest) -- not the timings, is what this covers."""
    from emu import build_emu, emu_engine
    if not __import__("os").path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++ to compile the kernels for the host")
    monkeypatch.setenv("RHIP_ARENA_PLACE_MIN_MB", "0")
    monkeypatch.setenv("RHIP_ARENA_TRIES", "3")
    monkeypatch.setenv("RHIP_ARENA_VMM", vmm)
    monkeypatch.setenv("RHIP_ARENA_VA_WINDOW_MB", "12")
    monkeypatch.setenv("RHIP_ARENA_VA_STEP_MB", "2")
    eng = emu_engine()
    try:
        G.arena_placement_body(eng, oracle, synth)
    finally:
        eng.close()


@pytest.mark.parametrize("gp8", ["0", "1"])
def test_emu_usmall_lds_variants(oracle, synth, monkeypatch, gp8):
    from emu import build_emu, emu_engine
    if not __import__("os").path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++ to compile the kernels for the host")
    monkeypatch.setenv("RHIP_USMALL_GP8", gp8)
    monkeypatch.setenv("RHIP_MERGE_CLASSES", "0")
    eng = emu_engine()
    try:
        G.usmall_variants_body(eng, oracle, synth)
    finally:
        eng.close()


def test_emu_batches_in_flight(emu, oracle, synth):
    G.test_batches_in_flight(emu, oracle, synth)


@pytest.mark.parametrize("mode", ["group", "groupfork"])
def test_emu_grouped_queues(oracle, synth, monkeypatch, mode):
    """The X-grouped image queues (k_count histogram -> scan -> k_emit bucket slots -> k_filter_g / k_union_g), forced
    on the small inputs with RHIP_GROUP_X=2; plus a sample of weather_sept_85 pairs against the reference fixture."""
    from emu import build_emu, emu_engine
    if not __import__("os").path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++ to compile the kernels for the host")
    monkeypatch.setenv("RHIP_GROUP_X", "2")
    if mode == "groupfork":
        monkeypatch.setenv("RHIP_FORK_MIN_MB", "0")
        monkeypatch.setenv("RHIP_SPIN_JOIN", "2")
    eng = emu_engine()
    try:
        G.grouped_body(eng, oracle, synth)
        if mode == "group":
            test_emu_realdata_pair_sample(eng, "weather_sept_85")
    finally:
        eng.close()
