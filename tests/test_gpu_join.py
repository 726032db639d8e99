"""The flag join of a forked batch (k_join_signal / k_join_wait in front of k_tail) and its fallback.  HIP does not promise
that kernels of different streams run side by side; when the gate gives up, the tail behind it has run too early, and
rhip_pairwise_end finishes the batch through events and a second tail -- the caller must get the same bytes either way
(the reference's functions cannot fail for scheduling reasons, roaring.h:102-113)."""
import threading

import numpy as np
import pytest

from util import all_pairs, load_bundle

pytestmark = pytest.mark.gpu

CHECKSUMS = {"and": 24220711, "or": 1232335437, "xor": 1208114726, "andnot": 581541349}  # SURVEY 8d, weather_sept_85


def _weather(eng):
    bufs = load_bundle("weather_sept_85")
    return eng.pool_from_serialized(bufs), len(bufs)


@pytest.mark.parametrize("mode", ["fail", "spins0"])
def test_join_fallback(monkeypatch, mode):
    """fail: every gate REPORTS a time-out (the fallback's bookkeeping: scratch reset, second tail, statistics);
    spins0: every gate gives up at once, so the first tail really runs before the class kernels have finished."""
    import croaring_amd
    monkeypatch.setenv("RHIP_SPIN_JOIN", "2")  # flags without the context's self-test
    if mode == "fail":
        monkeypatch.setenv("RHIP_JOIN_FAIL", "1")
    else:
        monkeypatch.setenv("RHIP_JOIN_SPINS", "0")
    eng = croaring_amd.Engine(0)
    ref_eng = None
    try:
        pool, n = _weather(eng)
        lhs, rhs = all_pairs(n)
        for op, want in CHECKSUMS.items():
            res = eng.pairwise(op, pool, lhs, pool, rhs)
            assert int(res.cardinalities().sum()) == want, op
        if mode == "fail":  # (spins0: whether a gate really gives up depends on which stream finishes first)
            assert eng.join_recovered() >= 1  # the first forked batch went through the fallback (later ones join with events)
        # bytes, against a context that never used flags
        monkeypatch.setenv("RHIP_SPIN_JOIN", "0")
        monkeypatch.delenv("RHIP_JOIN_FAIL", raising=False)
        monkeypatch.delenv("RHIP_JOIN_SPINS", raising=False)
        ref_eng = croaring_amd.Engine(0)
        rpool, _ = _weather(ref_eng)
        sub_l, sub_r = lhs[::37].copy(), rhs[::37].copy()
        monkeypatch.setenv("RHIP_SPIN_JOIN", "2")
        monkeypatch.setenv("RHIP_JOIN_SPINS", "0")
        monkeypatch.setenv("RHIP_JOIN_FAIL", "1")
        eng2 = croaring_amd.Engine(0)
        pool2, _ = _weather(eng2)
        a = eng2.pairwise("or", pool2, lhs, pool2, rhs)  # all pairs: a forked batch, through the fallback
        b = ref_eng.pairwise("or", rpool, lhs, rpool, rhs)
        ba, oa = a.serialize_many()
        bb, ob = b.serialize_many()
        assert np.array_equal(oa, ob) and np.array_equal(ba, bb)
        assert eng2.join_recovered() >= 1
        eng2.close()
    finally:
        eng.close()
        if ref_eng is not None:
            ref_eng.close()


def test_join_concurrent_contexts():
    """Eight threads, a context each, forked all-pairs batches side by side on one device -- every context's four streams
    share the hardware queues with the others'.  Whatever the gates see, every batch returns the reference's checksums."""
    import croaring_amd
    errs = []

    def work(tid):
        try:
            eng = croaring_amd.Engine(0)
            pool, n = _weather(eng)
            lhs, rhs = all_pairs(n)
            res = {op: None for op in CHECKSUMS}
            for it in range(25):
                for op, want in CHECKSUMS.items():
                    res[op] = eng.pairwise(op, pool, lhs, pool, rhs, reuse=res[op])
                    got = int(res[op].cardinalities().sum())
                    if got != want:
                        errs.append((tid, it, op, got, want))
                        return
            eng.close()
        except Exception as e:  # noqa
            errs.append((tid, repr(e)))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not errs, errs[:3]
