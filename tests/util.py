"""Shared helpers for the test-suite: fixture loading, checksums."""
from __future__ import annotations

import lzma
import os
import struct
import zlib

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OPS = ("and", "or", "xor", "andnot")
DATASETS = ("census1881", "weather_sept_85", "wikileaks-noquotes", "census-income")


def load_bundle(name: str) -> list:
    """tests/golden/<name>.rbnd.xz -> list of portable-serialized bitmaps."""
    with lzma.open(os.path.join(GOLD, f"{name}.rbnd.xz"), "rb") as f:
        raw = f.read()
    assert raw[:4] == b"RBND"
    n = struct.unpack_from("<I", raw, 4)[0]
    out, p = [], 8
    for _ in range(n):
        ln = struct.unpack_from("<I", raw, p)[0]
        out.append(raw[p + 4:p + 4 + ln])
        p += 4 + ln
    return out


def load_pairs(name: str):
    return np.load(os.path.join(GOLD, f"{name}_pairs.npz"))


def crc(b: bytes) -> int:
    return zlib.crc32(b) & 0xFFFFFFFF


def all_pairs(n: int):
    i, j = np.triu_indices(n, 1)
    return i.astype(np.uint32), j.astype(np.uint32)


def synth_inputs():
    """The seeded synthetic inputs of oracle/gen_golden.py (same code path, same seed)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(root, "oracle", "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gg)
    return gg.synth_inputs()


def c5_inputs(base=None) -> list:
    """wikileaks-noquotes replicated into 10 high-32 buckets, v + (r << 32), as roaring64 portable
    images (src/roaring64.c:2323-2393: u64 bucket count, then per bucket u32 high + the 32-bit portable image)."""
    base = load_bundle("wikileaks-noquotes") if base is None else base
    return [struct.pack("<Q", 10) + b"".join(struct.pack("<I", r) + b for r in range(10)) for b in base]
