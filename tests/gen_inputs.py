"""Seeded synthetic bitmap generators shared by the parity tests (no reference needed).

Every generator returns sorted uint32 numpy arrays; the test turns them into
portable-serialized bitmaps through the oracle (``from_sorted`` + run_optimize),
so the same bytes feed the oracle, the reference (when present) and the HIP engine.
"""
from __future__ import annotations

import numpy as np

PROFILES = ("sparse", "tiny", "mid", "dense", "verydense", "runs", "shortruns", "full",
            "nearfull", "single", "edge", "boundary4096", "boundary4097", "blocks")


def chunk_values(rng: np.random.Generator, profile: str) -> np.ndarray:
    """Low-16-bit values (sorted, unique) for one container of the given profile."""
    if profile == "sparse":
        n = int(rng.integers(1, 600))
        return np.unique(rng.integers(0, 65536, n))
    if profile == "tiny":
        n = int(rng.integers(1, 40))
        return np.unique(rng.integers(0, 65536, n))
    if profile == "mid":
        n = int(rng.integers(2000, 4096))
        return np.unique(rng.integers(0, 65536, n))
    if profile == "dense":
        return np.flatnonzero(rng.random(65536) < 0.5)
    if profile == "verydense":
        return np.flatnonzero(rng.random(65536) < 0.97)
    if profile == "runs":
        k = int(rng.integers(1, 12))
        cuts = np.sort(rng.choice(65536, 2 * k, replace=False))
        return np.unique(np.concatenate([np.arange(cuts[2 * i], cuts[2 * i + 1]) for i in range(k)] + [cuts[:1]]))
    if profile == "shortruns":
        k = int(rng.integers(20, 900))
        starts = np.sort(rng.choice(65000, k, replace=False))
        lens = rng.integers(1, 40, k)
        return np.unique(np.concatenate([np.arange(s, min(65536, s + l)) for s, l in zip(starts, lens)]))
    if profile == "full":
        return np.arange(65536)
    if profile == "nearfull":
        v = np.ones(65536, bool)
        v[rng.integers(0, 65536, int(rng.integers(1, 30)))] = False
        return np.flatnonzero(v)
    if profile == "single":
        return np.array([int(rng.integers(0, 65536))])
    if profile == "edge":
        return np.unique(np.concatenate([[0, 65535], rng.integers(0, 65536, 5), [63, 64, 127, 128]]))
    if profile == "boundary4096":
        return np.sort(rng.choice(65536, 4096, replace=False))
    if profile == "boundary4097":
        return np.sort(rng.choice(65536, 4097, replace=False))
    if profile == "blocks":
        v = np.zeros(65536, bool)
        for s in rng.integers(0, 1024, int(rng.integers(1, 200))):
            v[s * 64:(s + 1) * 64] = True
        return np.flatnonzero(v)
    raise ValueError(profile)


def random_bitmap(rng: np.random.Generator, max_keys: int = 12, key_space: int = 24,
                  profiles=PROFILES) -> np.ndarray:
    nk = int(rng.integers(0, max_keys + 1))
    keys = np.sort(rng.choice(key_space, min(nk, key_space), replace=False))
    parts = []
    for k in keys:
        p = profiles[int(rng.integers(0, len(profiles)))]
        parts.append((np.uint32(k) << np.uint32(16)) | chunk_values(rng, p).astype(np.uint32))
    if not parts:
        return np.zeros(0, np.uint32)
    return np.concatenate(parts).astype(np.uint32)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n outputs of splitmix64 (word i of bitmap b = splitmix64(seed_b) stream)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))
