"""Generate tests/golden/* with the REAL reference (oracle/_ref, built from /root/reference).

Run in the build container only (needs /root/reference for the realdata text files):
    python oracle/gen_golden.py

Outputs (committed; the GPU box cannot see /root/reference):
  tests/golden/<dataset>.rbnd.xz    portable-serialized input bitmaps, built exactly as the
                                    reference benchmark does (benchmarks/benchmark.cpp:1938-1942:
                                    roaring_bitmap_of_ptr + run_optimize + shrink_to_fit), files in
                                    alphasort order (benchmarks/numbersfromtextfiles.h:113-118).
  tests/golden/<dataset>_pairs.npz  for ALL unordered pairs i<j and each op in (and, or, xor, andnot):
                                    result cardinality, portable size and crc32 of the reference's
                                    portable serialization; plus or_many / or_many_heap / xor_many
                                    results (serialized) over the whole dataset.
  tests/golden/bitmapwith{,out}runs.bin   Java-produced format fixtures, verbatim copies of
                                    tests/testdata/* (format_portability_unit.c:51-90).
  tests/golden/c5_wikileaks64_pairs.npz  BASELINE config C5: roaring64, wikileaks-noquotes x 10 high-32 buckets -- all pairs x
                                    4 ops (cardinality, portable size, crc32) + the 200-way union (serialized).
  tests/golden/c4x10_or_many.npz    the C4 generator at 10^6 bitmaps (opt-in: `gen_golden.py c4x10`): cardinality / size / crc32
  tests/golden/c4_or_many.npz       BASELINE config C4: roaring_bitmap_or_many over the 100 000 seeded sparse bitmaps
                                    (cardinality, size, crc32 of the reference's result; crc32 of the inputs).
  tests/golden/frozen.npz           (`gen_golden.py frozen`) size and crc32 of roaring_bitmap_frozen_serialize's image of every
                                    bitmap of the four realdata bundles and of the seeded synthetic inputs
  tests/golden/frozen_withruns.bin  the reference's frozen image of bitmapwithruns.bin (a verbatim reader fixture)
  tests/golden/synth_mixed.npz      crc32/size/cardinality of the reference's results on seeded synthetic
                                    bitmaps hitting every container-type pair and result-typing branch
                                    (inputs are regenerated from the seed; their crc32 is pinned too).
"""
import lzma
import os
import shutil
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pyoracle import Ref  # noqa: E402
from gen_inputs import random_bitmap, PROFILES, chunk_values  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
OPS = ("and", "or", "xor", "andnot")


def write_bundle(path, blobs):
    raw = b"RBND" + struct.pack("<I", len(blobs)) + b"".join(struct.pack("<I", len(b)) + b for b in blobs)
    with lzma.open(path, "wb", preset=9) as f:
        f.write(raw)


def load_text_dataset(R, name):
    d = os.path.join(REF, "benchmarks", "realdata", name)
    files = sorted(f for f in os.listdir(d) if f.endswith(".txt"))  # alphasort == byte order
    hs = []
    for f in files:
        txt = open(os.path.join(d, f)).read().strip()
        vals = np.array(sorted(set(int(x) for x in txt.split(",") if x.strip())), dtype=np.uint32)
        hs.append(R.from_sorted(vals))
    return hs


def pairs_golden(R, hs, name):
    n = len(hs)
    idx = [(i, j) for i in range(n) for j in range(i + 1, n)]
    out = {}
    for op in OPS:
        card = np.zeros(len(idx), np.uint32)
        size = np.zeros(len(idx), np.uint32)
        crc = np.zeros(len(idx), np.uint32)
        for k, (i, j) in enumerate(idx):
            r = R.op(op, hs[i], hs[j])
            s = R.serialize(r)
            card[k] = R.cardinality(r)
            size[k] = len(s)
            crc[k] = zlib.crc32(s)
            assert R.op_cardinality(op, hs[i], hs[j]) == card[k]
            R.free(r)
        out[f"{op}_card"], out[f"{op}_size"], out[f"{op}_crc"] = card, size, crc
        print(f"  {name} {op}: sum card = {int(card.astype(np.uint64).sum())}")
    for nm, fn in (("or_many", R.or_many), ("or_many_heap", R.or_many_heap), ("xor_many", R.xor_many)):
        r = fn(hs)
        out[nm] = np.frombuffer(R.serialize(r), dtype=np.uint8)
        print(f"  {name} {nm}: card = {R.cardinality(r)}")
        R.free(r)
    out["pairs"] = np.array(idx, dtype=np.uint16)
    np.savez_compressed(os.path.join(GOLD, f"{name}_pairs.npz"), **out)


def synth_inputs():
    """Seeded synthetic inputs (sorted uint32 arrays).  Shared with tests/test_golden_*.py,
    which regenerate them and check their crc32 against the fixture before using them."""
    rng = np.random.default_rng(20260923)
    singles = []
    for pa in PROFILES:
        for pb in PROFILES:
            a = (np.uint32(5) << np.uint32(16)) | chunk_values(rng, pa).astype(np.uint32)
            b = (np.uint32(5) << np.uint32(16)) | chunk_values(rng, pb).astype(np.uint32)
            singles.append((a, b))
    for _ in range(120):
        singles.append((random_bitmap(rng), random_bitmap(rng)))
    many = []
    for _ in range(40):
        k = int(rng.integers(0, 10))
        many.append([random_bitmap(rng, max_keys=6, key_space=8) for _ in range(k)])
    return singles, many


def synth_golden(R):
    """Every (profile_a, profile_b) container pair under one key + random multi-key bitmaps.
    Only crc32/size/cardinality are stored: inputs are regenerated from the seed by the tests."""
    singles, many = synth_inputs()
    out = {"in_crc": [], "in_size": []}
    for op in OPS:
        out[f"{op}_crc"], out[f"{op}_size"], out[f"{op}_card"] = [], [], []
    for a, b in singles:
        ha, hb = R.from_sorted(a), R.from_sorted(b)
        for h in (ha, hb):
            s = R.serialize(h)
            out["in_crc"].append(zlib.crc32(s)); out["in_size"].append(len(s))
        for op in OPS:
            r = R.op(op, ha, hb)
            s = R.serialize(r)
            out[f"{op}_crc"].append(zlib.crc32(s)); out[f"{op}_size"].append(len(s))
            out[f"{op}_card"].append(R.cardinality(r))
            R.free(r)
        R.free(ha); R.free(hb)
    for nm in ("or_many", "xor_many", "or_many_heap"):
        out[f"{nm}_crc"], out[f"{nm}_size"], out[f"{nm}_card"] = [], [], []
    out["many_in_crc"] = []
    for vs in many:
        hs = [R.from_sorted(v) for v in vs]
        out["many_in_crc"].append(zlib.crc32(b"".join(R.serialize(h) for h in hs)))
        for nm, fn in (("or_many", R.or_many), ("xor_many", R.xor_many), ("or_many_heap", R.or_many_heap)):
            r = fn(hs)
            s = R.serialize(r)
            out[f"{nm}_crc"].append(zlib.crc32(s)); out[f"{nm}_size"].append(len(s))
            out[f"{nm}_card"].append(R.cardinality(r))
            R.free(r)
        for h in hs:
            R.free(h)
    np.savez_compressed(os.path.join(GOLD, "synth_mixed.npz"),
                        **{k: np.array(v, dtype=np.uint32) for k, v in out.items()})
    print(f"  synth: {len(singles)} pairs, {len(many)} many-way groups")


def c5_golden(R):
    """All unordered pairs x 4 ops through the real roaring64_bitmap_{and,or,xor,andnot} (roaring64.c:1332-1373,
    1541-1593, 1663-1720, 1809-1861) + the 200-way union as a left fold of roaring64_bitmap_or_inplace."""
    from util import c5_inputs, load_bundle
    base = load_bundle("wikileaks-noquotes")
    bufs = c5_inputs(base)
    hs = [R.deserialize64(b) for b in bufs]
    # the constructed images ARE what the reference writes for these bitmaps (pins the construction), and the
    # bitmaps are what inserting v + (r << 32) + run_optimize gives
    for k in (0, 57, 199):
        assert R.serialize64(hs[k]) == bufs[k]
        h32 = R.deserialize(base[k])
        v = R.to_array(h32).astype(np.uint64)
        R.free(h32)
        vals = np.sort(np.concatenate([v + (np.uint64(r) << np.uint64(32)) for r in range(10)]))
        hv = R.from_sorted64(vals)
        assert R.serialize64(hv) == bufs[k]
        R.free64(hv)
    n = len(hs)
    idx = [(i, j) for i in range(n) for j in range(i + 1, n)]
    out = {"in_crc": np.array([zlib.crc32(b) for b in bufs], np.uint32)}
    for op in OPS:
        card = np.zeros(len(idx), np.uint64)
        size = np.zeros(len(idx), np.uint32)
        crc = np.zeros(len(idx), np.uint32)
        for k, (i, j) in enumerate(idx):
            r = R.op64(op, hs[i], hs[j])
            s = R.serialize64(r)
            card[k], size[k], crc[k] = R.cardinality64(r), len(s), zlib.crc32(s)
            R.free64(r)
        out[f"{op}_card"], out[f"{op}_size"], out[f"{op}_crc"] = card, size, crc
        print(f"  c5 {op}: sum card = {int(card.sum())}")
    r = R.or_many64(hs)
    out["fold_or"] = np.frombuffer(R.serialize64(r), dtype=np.uint8)
    out["fold_or_card"] = np.array([R.cardinality64(r)], np.uint64)
    print(f"  c5 200-way union: card = {R.cardinality64(r)}, {out['fold_or'].size} bytes")
    R.free64(r)
    out["pairs"] = np.array(idx, dtype=np.uint16)
    np.savez_compressed(os.path.join(GOLD, "c5_wikileaks64_pairs.npz"), **out)
    for h in hs:
        R.free64(h)


def c4_golden(R, n_bitmaps=100000):
    """SURVEY §8d C4 at FULL size: roaring_bitmap_or_many (roaring.c:775-790) over the 100 000 seeded sparse bitmaps
    of rhip_synth_sparse_sizes/_fill (the generator's definition is in include/roaring_hip.h); also the G-way
    sharded variants' common answer is this one.  Stored: crc32 of the input blob, cardinality / size / crc32 of the
    reference's serialized result, and the same for the first 10 000 and first 1 000 bitmaps."""
    import croaring_amd
    blob, offs = croaring_amd.synth_sparse_portable(0, 1, n_bitmaps)
    mv = memoryview(blob)
    hs = [R.deserialize(bytes(mv[int(offs[b]):int(offs[b + 1])])) for b in range(n_bitmaps)]
    assert all(R.validate(h) for h in hs[:64])
    out = {"n_bitmaps": np.array([n_bitmaps], np.uint64), "in_crc": np.array([zlib.crc32(blob)], np.uint32),
           "in_bytes": np.array([blob.size], np.uint64)}
    for n in (1000, 10000, n_bitmaps):
        r = R.or_many(hs[:n])
        s = R.serialize(r)
        out[f"or_many_{n}"] = np.array([R.cardinality(r), len(s), zlib.crc32(s)], np.uint64)
        print(f"  c4 or_many over {n}: card {R.cardinality(r)}, {len(s)} bytes, crc {zlib.crc32(s)}")
        R.free(r)
    r = R.xor_many(hs[:1000])
    out["xor_many_1000_card"] = np.array([R.cardinality(r)], np.uint64)
    R.free(r)
    np.savez_compressed(os.path.join(GOLD, "c4_or_many.npz"), **out)
    for h in hs:
        R.free(h)


def c4x10_golden(R, n_bitmaps=1000000, chunk=50000):
    """The C4 generator at 10 x the size (10^6 bitmaps, 16.6 GB of portable images): the row of the bench where the
    many-way reduction is large enough for N GPUs to split it.  The reference's answer, chunk by chunk:
    roaring_bitmap_or_many over `chunk` bitmaps at a time, the partial unions folded with roaring_bitmap_or (set-equal
    to one or_many over all of them).  Stored: cardinality, container count and crc32 of the serialized result."""
    import croaring_amd
    acc = None
    for first in range(0, n_bitmaps, chunk):
        cnt = min(chunk, n_bitmaps - first)
        blob, offs = croaring_amd.synth_sparse_portable(first, 1, cnt)
        mv = memoryview(blob)
        hs = [R.deserialize(bytes(mv[int(offs[b]):int(offs[b + 1])])) for b in range(cnt)]
        part = R.or_many(hs)
        for h in hs:
            R.free(h)
        if acc is None:
            acc = part
        else:
            nxt = R.op("or", acc, part)
            R.free(acc)
            R.free(part)
            acc = nxt
        print(f"  c4x10: {first + cnt} bitmaps, card {R.cardinality(acc)}", flush=True)
    R.run_optimize(acc)
    s = R.serialize(acc)
    out = {"n_bitmaps": np.array([n_bitmaps], np.uint64),
           "or_many": np.array([R.cardinality(acc), len(s), zlib.crc32(s)], np.uint64)}
    np.savez_compressed(os.path.join(GOLD, "c4x10_or_many.npz"), **out)
    print("  c4x10 or_many:", out["or_many"])
    R.free(acc)


def read_bundle(path):
    raw = lzma.open(path, "rb").read()
    n = struct.unpack_from("<I", raw, 4)[0]
    out, p = [], 8
    for _ in range(n):
        ln = struct.unpack_from("<I", raw, p)[0]
        out.append(raw[p + 4:p + 4 + ln])
        p += 4 + ln
    return out


def frozen_golden(R):
    """roaring_bitmap_frozen_serialize (src/roaring.c:3242-3328) of every realdata bitmap (inputs: the committed portable
    bundles) and of the seeded synthetic inputs: size + crc32; one image verbatim."""
    out = {}
    for name in ("census1881", "weather_sept_85", "wikileaks-noquotes", "census-income"):
        bufs = read_bundle(os.path.join(GOLD, f"{name}.rbnd.xz"))
        crc, size = [], []
        for b in bufs:
            h = R.deserialize(b)
            f = R.frozen_serialize(h)
            crc.append(zlib.crc32(f)); size.append(len(f))
            R.free(h)
        out[f"{name}_crc"], out[f"{name}_size"] = crc, size
    singles, _ = synth_inputs()
    crc, size = [], []
    for a, b in singles:
        for v in (a, b):
            h = R.from_sorted(v)
            f = R.frozen_serialize(h)
            crc.append(zlib.crc32(f)); size.append(len(f))
            R.free(h)
    out["synth_crc"], out["synth_size"] = crc, size
    np.savez_compressed(os.path.join(GOLD, "frozen.npz"), **{k: np.array(v, dtype=np.uint32) for k, v in out.items()})
    h = R.deserialize(open(os.path.join(GOLD, "bitmapwithruns.bin"), "rb").read())
    open(os.path.join(GOLD, "frozen_withruns.bin"), "wb").write(R.frozen_serialize(h))
    R.free(h)
    print("  frozen:", {k: len(v) for k, v in out.items() if k.endswith("_crc")})


def robust_corpus():
    """The malformed / borderline portable images the reference's own tests hold
    (tests/robust_deserialization_unit.c:98-124 crashproneinput1-7.bin, :168-388 the hand-made vectors;
    tests/cpp_roaring64_unit.cpp:427-439 the four bad 64map*.bin) as (name, bytes, is64)."""
    td = os.path.join(REF, "tests", "testdata")
    out = [(f"crashproneinput{i}", open(os.path.join(td, f"crashproneinput{i}.bin"), "rb").read(), 0) for i in range(1, 8)]
    b = bytes
    out.append(("negative_container_count", b([0x3A, 0x30, 0, 0, 0, 0, 0, 0x80]), 0))
    n = (1 << 16)
    out.append(("max_container_count_valid", b([0x3A, 0x30, 0, 0, 0, 0, 0x01, 0]) + b(n * 10), 0))
    out.append(("huge_container_count", b([0x3A, 0x30, 0, 0, 1, 0, 0x01, 0]) + b((n + 1) * 10), 0))
    hdr = [0x3B, 0x30, 0, 0]
    out.append(("run_container_empty", b(hdr + [1, 0, 0, 0, 0, 0, 0]), 0))
    out.append(("run_should_combine", b(hdr + [1, 0, 0, 1, 0, 2, 0, 0, 0, 0, 0, 1, 0, 0, 0]), 0))
    out.append(("run_overlap", b(hdr + [1, 0, 0, 4, 0, 2, 0, 0, 0, 4, 0, 1, 0, 0, 0]), 0))
    out.append(("run_overflow", b(hdr + [1, 0, 0, 4, 0, 1, 0, 0xFE, 0xFF, 4, 0]), 0))
    out.append(("run_incorrect_cardinality_still_allowed", b(hdr + [1, 0, 0, 0, 0, 1, 0, 0, 0, 8, 0]), 0))
    two = [0x3B, 0x30, 1, 0, 0]
    out.append(("duplicate_keys", b(two + [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]), 0))
    out.append(("unsorted_keys", b(two + [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]), 0))
    out.append(("unsorted_array", b(hdr + [0, 0, 0, 1, 0, 1, 0, 0, 0]), 0))
    out.append(("duplicate_array", b(hdr + [0, 0, 0, 1, 0, 0, 0, 0, 0]), 0))
    full = b(hdr + [0, 0, 0, 0xFF, 0xFF]) + b([0xFF]) * 8192
    out.append(("bitset_full_valid", full, 0))
    out.append(("bitset_incorrect_cardinality", full[:-1] + b([0xFE]), 0))
    for i in range(len(out)):  # truncated by one byte: never a bitmap (robust_deserialization_unit.c:147-149)
        nm, d, _ = out[i]
        if not nm.startswith("crash") and len(d) < 100000:
            out.append((nm + "_truncated", d[:-1], 0))
    for f in ("64mapemptyinput", "64mapsizetoosmall", "64mapinvalidsize", "64mapkeytoosmall",
              "64map32bitvals", "64mapspreadvals", "64maphighvals", "64mapempty"):
        out.append((f, open(os.path.join(td, f + ".bin"), "rb").read(), 1))
    return out


def robust_golden(R):
    """For every vector: does the reference hand back a VALID bitmap (deserialize_safe non-NULL AND
    roaring_bitmap_internal_validate -- what robust_deserialization_unit.c:131-166 asserts), and its re-serialization."""
    import ctypes as C
    R.L.roaring64_bitmap_internal_validate.restype = C.c_bool
    R.L.roaring64_bitmap_internal_validate.argtypes = [C.c_void_p, C.POINTER(C.c_char_p)]
    names, blobs, is64, accept, nonnull, reser = [], [], [], [], [], []
    for nm, d, w in robust_corpus():
        reason = C.c_char_p()
        if w:
            h = R.L.roaring64_bitmap_portable_deserialize_safe(d, len(d))
            ok = bool(h) and bool(R.L.roaring64_bitmap_internal_validate(h, C.byref(reason)))
            ser = R.serialize64(h) if ok else b""
            if h:
                R.free64(h)
        else:
            h = R.L.roaring_bitmap_portable_deserialize_safe(d, len(d))
            ok = bool(h) and R.validate(h)
            ser = R.serialize(h) if ok else b""
            if h:
                R.free(h)
        names.append(nm); blobs.append(d); is64.append(w); accept.append(int(ok)); nonnull.append(int(bool(h))); reser.append(ser)
        print(f"  robust {nm}: {len(d)} bytes, reference {'accepts' if ok else ('returns an INVALID bitmap' if h else 'rejects')}")
    lens = np.array([len(d) for d in blobs], np.uint64)
    rlens = np.array([len(d) for d in reser], np.uint64)
    np.savez_compressed(os.path.join(GOLD, "robust_corpus.npz"), names=np.array(names), is64=np.array(is64, np.uint8),
                        accept=np.array(accept, np.uint8), nonnull=np.array(nonnull, np.uint8), lens=lens,
                        blob=np.frombuffer(b"".join(blobs), np.uint8), rlens=rlens,
                        reser=np.frombuffer(b"".join(reser), np.uint8))


def main():
    R = Ref()
    os.makedirs(GOLD, exist_ok=True)
    for f in ("bitmapwithruns.bin", "bitmapwithoutruns.bin"):
        shutil.copy(os.path.join(REF, "tests", "testdata", f), os.path.join(GOLD, f))
    for f in ("64map32bitvals.bin", "64mapspreadvals.bin", "64maphighvals.bin", "64mapempty.bin"):
        shutil.copy(os.path.join(REF, "tests", "testdata", f), os.path.join(GOLD, f))
    names = sys.argv[1:] or ["synth", "census1881", "weather_sept_85", "wikileaks-noquotes", "census-income", "c5", "c4"]
    if "synth" in names:
        synth_golden(R)
    if "c5" in names:
        c5_golden(R)
    if "c4" in names:
        c4_golden(R)
    if "frozen" in names:  # (reads the committed bundles: run after the datasets)
        frozen_golden(R)
    if "robust" in names:
        robust_golden(R)
    if "c4x10" in names:  # (not in the default list: 10^6 bitmaps, a few minutes)
        c4x10_golden(R)
    for name in [n for n in names if n not in ("synth", "c4", "c5", "c4x10", "frozen", "robust")]:
        print(name)
        hs = load_text_dataset(R, name)
        write_bundle(os.path.join(GOLD, f"{name}.rbnd.xz"), [R.serialize(h) for h in hs])
        pairs_golden(R, hs, name)
        for h in hs:
            R.free(h)


if __name__ == "__main__":
    main()
