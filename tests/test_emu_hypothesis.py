"""Property-based differential testing of the kernels through the CPU emulator (tests/emu): hypothesis builds pairs
of bitmaps out of the shapes that sit on container / typing boundaries (single values, dense ranges, runs ending at
65535 / starting at 0, full containers, 4096 / 4097-value arrays) and every op, flip, run_optimize and the value-list
round trip is compared with the oracle at byte level.  Deterministic (derandomize=True): the same examples every run."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from util import OPS

KEYS = st.integers(0, 5)
EDGE16 = st.sampled_from([0, 1, 2, 31, 32, 63, 64, 4095, 4096, 4097, 32767, 32768, 65534, 65535])
V16 = st.one_of(EDGE16, st.integers(0, 65535))


@st.composite
def chunk(draw):
    """Low-16 values of one container."""
    kind = draw(st.sampled_from(["few", "range", "runs", "full", "stride", "boundary", "manyruns", "midarray"]))
    if kind == "manyruns":   # 8 .. 140 short runs: the interval kernel's size classes (31 / 127 / 255 intervals a side)
        n = draw(st.sampled_from([8, 30, 31, 32, 33, 60, 126, 127, 128, 129, 140]))
        gap = draw(st.sampled_from([3, 5, 40, 400]))
        ln = draw(st.sampled_from([1, 2, 3, 9]))
        s0 = draw(st.integers(0, 2000))
        starts = s0 + np.arange(n, dtype=np.uint32) * np.uint32(gap + ln)
        v = (starts[:, None] + np.arange(ln, dtype=np.uint32)[None, :]).ravel()
        return np.unique(v[v < 65536]).astype(np.uint32)
    if kind == "midarray":   # 100 .. 300 scattered values: the probe / rank-merge limits (128 / 255 / 256 values)
        n = draw(st.sampled_from([100, 127, 128, 129, 200, 254, 255, 256, 257, 300]))
        rng = np.random.default_rng(draw(st.integers(0, 2 ** 31)))
        lo = draw(st.sampled_from([0, 30000, 65536 - 4 * 300]))
        return np.sort(rng.choice(4 * 300, n, replace=False).astype(np.uint32) + np.uint32(lo))
    if kind == "few":
        return np.unique(np.array(draw(st.lists(V16, min_size=1, max_size=12)), dtype=np.uint32))
    if kind == "range":
        a, b = sorted((draw(V16), draw(V16)))
        return np.arange(a, b + 1, dtype=np.uint32)
    if kind == "runs":
        parts = []
        for _ in range(draw(st.integers(1, 6))):
            a = draw(V16)
            n = draw(st.sampled_from([1, 2, 3, 17, 300, 5000]))
            parts.append(np.arange(a, min(a + n, 65536), dtype=np.uint32))
        return np.unique(np.concatenate(parts))
    if kind == "full":
        return np.arange(0, 65536, dtype=np.uint32)
    if kind == "stride":
        step = draw(st.sampled_from([2, 3, 7, 15, 16, 17]))
        return np.arange(draw(st.integers(0, step - 1)), 65536, step, dtype=np.uint32)
    n = draw(st.sampled_from([4095, 4096, 4097]))           # array / bitset boundary
    off = draw(st.sampled_from([0, 1, 61439]))
    return (np.arange(n, dtype=np.uint32) + off) % 65536 if off else np.arange(n, dtype=np.uint32) * 2 % 65536


@st.composite
def bitmap(draw):
    keys = sorted(set(draw(st.lists(KEYS, min_size=0, max_size=4))))
    parts = [(np.uint32(k) << np.uint32(16)) | np.unique(draw(chunk())) for k in keys]
    v = np.unique(np.concatenate(parts)) if parts else np.zeros(0, np.uint32)
    return v.astype(np.uint32), draw(st.booleans())


@pytest.fixture(scope="module")
def emu():
    from emu import build_emu, emu_engine
    if not __import__("os").path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++ to compile the kernels for the host")
    eng = emu_engine()
    yield eng
    eng.close()


# RHIP_HYP_EXAMPLES=n / RHIP_HYP_RANDOM=1: a longer, randomised hunt (offline); the committed default is deterministic
_os = __import__("os")
CFG = dict(max_examples=int(_os.environ.get("RHIP_HYP_EXAMPLES", "120")), deadline=None,
           derandomize=_os.environ.get("RHIP_HYP_RANDOM") != "1", database=None, suppress_health_check=list(HealthCheck))


@settings(**CFG)
@given(a=bitmap(), b=bitmap())
def test_pairwise_ops_match_the_oracle(emu, oracle, a, b):
    (va, ra), (vb, rb) = a, b
    ha, hb = oracle.from_sorted(va, run_optimize=ra), oracle.from_sorted(vb, run_optimize=rb)
    pool = emu.pool_from_serialized([oracle.serialize(ha), oracle.serialize(hb)])
    for op in OPS:
        for l, r, x, y in ((0, 1, ha, hb), (1, 0, hb, ha), (0, 0, ha, ha)):
            want = oracle.op(op, x, y)
            got = emu.pairwise(op, pool, [l], pool, [r])
            assert got.serialize(0) == oracle.serialize(want), (op, l, r)
            assert emu.pairwise_cardinality(op, pool, [l], pool, [r])[0] == oracle.cardinality(want)
            oracle.free(want)
    for nm, fn, of in (("or", emu.or_many, oracle.or_many), ("xor", emu.xor_many, oracle.xor_many)):
        want = of([ha, hb])
        g = oracle.deserialize(fn(pool).serialize(0))
        assert np.array_equal(oracle.to_array(g), oracle.to_array(want)), nm
        oracle.free(want)
        oracle.free(g)
    oracle.free(ha)
    oracle.free(hb)


@settings(**CFG)
@given(a=bitmap(), s=st.integers(0, (6 << 16) + 10), n=st.sampled_from([0, 1, 2, 3, 65535, 65536, 65537, 200000, 1 << 32]))
def test_flip_convert_and_value_lists_match_the_oracle(emu, oracle, a, s, n):
    va, ra = a
    h = oracle.from_sorted(va, run_optimize=ra)
    pool = emu.pool_from_serialized([oracle.serialize(h)])
    want = oracle.flip(h, s, s + n)
    assert emu.flip(pool, [s], [s + n]).serialize(0) == oracle.serialize(want)
    oracle.free(want)
    built = emu.pool_from_values([va])
    plain = oracle.from_sorted(va, run_optimize=False)
    assert built.serialize(0) == oracle.serialize(plain)
    oracle.run_optimize(plain)
    assert emu.run_optimize(built).serialize(0) == oracle.serialize(plain)
    oracle.remove_run_compression(plain)
    assert emu.remove_run_compression(pool).serialize(0) == oracle.serialize(plain)
    vals, offs = pool.to_values()
    assert np.array_equal(vals, va)
    blob, boffs = pool.serialize_many()
    assert emu.pool_from_blob(blob, boffs).serialize(0) == pool.serialize(0)
    # the frozen format both ways (roaring_bitmap_frozen_serialize / _view): the oracle's bytes, and back
    fblob, foffs, flens = pool.frozen_serialize_many()
    assert fblob[int(foffs[0]):int(foffs[0]) + int(flens[0])].tobytes() == oracle.frozen_serialize(h)
    assert emu.pool_from_frozen(fblob, foffs, flens).serialize(0) == pool.serialize(0)
    oracle.free(plain)
    oracle.free(h)


@settings(**{**CFG, "max_examples": max(20, CFG["max_examples"] // 4)})
@given(bms=st.lists(bitmap(), min_size=3, max_size=7), op=st.sampled_from(OPS))
def test_batched_all_pairs_match_the_oracle(emu, oracle, bms, op):
    """Several bitmaps, every ordered pair in ONE call: the sub-wave groups of the planning, interval and copy kernels
    then hold different items (sizes, types, empty results) in one wave."""
    hs = [oracle.from_sorted(v, run_optimize=r) for v, r in bms]
    pool = emu.pool_from_serialized([oracle.serialize(h) for h in hs])
    n = len(hs)
    lhs = np.repeat(np.arange(n, dtype=np.uint32), n)
    rhs = np.tile(np.arange(n, dtype=np.uint32), n)
    res = emu.pairwise(op, pool, lhs, pool, rhs)
    cards = emu.pairwise_cardinality(op, pool, lhs, pool, rhs)
    # the same batch over a PREPARED pair list (rhip_pairlist_*): no host pass, no staging -- identical bytes
    plist = emu.pairlist(pool, lhs, pool, rhs)
    assert np.array_equal(emu.pairwise_list(op, plist).serialize_many()[0], res.serialize_many()[0])
    assert np.array_equal(emu.pairwise_list_cardinality(op, plist), cards)
    plist.free()
    for k in range(lhs.size):
        want = oracle.op(op, hs[lhs[k]], hs[rhs[k]])
        assert res.serialize(k) == oracle.serialize(want), (op, int(lhs[k]), int(rhs[k]))
        assert cards[k] == oracle.cardinality(want)
        oracle.free(want)
    for h in hs:
        oracle.free(h)
