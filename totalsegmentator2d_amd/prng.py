"""Portable counter-based PRNG for synthetic weights and inputs.

185 MB of weights cannot be committed and the real Zenodo checkpoints (reference ``ts2d/data/shared.json:1-34``) are
a run-time download, so every test / bench tensor is regenerated from ``(seed, stream, index)``.  Only exact
integer arithmetic and IEEE double multiply/subtract are used (no libm transcendental), so the same bits come
out on any host: splitmix64 hashing of the counter, eight 16-bit uniforms summed (Irwin-Hall, n=8) and
standardised -> approximately N(0,1) with support +-4.9 sigma.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_IH_SCALE = 1.0 / (65536.0 * (8.0 / 12.0) ** 0.5)   # std of a sum of 8 U{0..65535} is 65536*sqrt(8/12) (to 1e-10)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over='ignore'):
        x = (x + _GOLDEN) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _key(seed: int, stream: int) -> np.uint64:
    k = _splitmix64(np.array([seed & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))
    k = _splitmix64(k ^ np.uint64(stream & 0xFFFFFFFFFFFFFFFF))
    return k[0]


def key(seed: int, stream: int) -> int:
    """The 64-bit stream key as a Python int (what the device generator ts2d_synth_slices takes)."""
    return int(_key(seed, stream))


def hash_u64(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n 64-bit words for counters offset..offset+n-1."""
    with np.errstate(over='ignore'):
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        return _splitmix64((idx * np.uint64(0xD1342543DE82EF95) + _key(seed, stream)) & _M64)


def normal(seed: int, stream: int, n: int, offset: int = 0, chunk: int = 1 << 22) -> np.ndarray:
    """n approximately-N(0,1) float64 values; value i depends only on (seed, stream, offset+i)."""
    out = np.empty(n, dtype=np.float64)
    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        # two words per value: counters 2i and 2i+1
        with np.errstate(over='ignore'):
            idx = np.arange(offset + a, offset + a + m, dtype=np.uint64)
            k = _key(seed, stream)
            h1 = _splitmix64(((idx * np.uint64(2)) * np.uint64(0xD1342543DE82EF95) + k) & _M64)
            h2 = _splitmix64(((idx * np.uint64(2) + np.uint64(1)) * np.uint64(0xD1342543DE82EF95) + k) & _M64)
        s = np.zeros(m, dtype=np.int64)
        for sh in (0, 16, 32, 48):
            s += ((h1 >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
            s += ((h2 >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
        out[a:a + m] = (s - 4 * 65535).astype(np.float64) * _IH_SCALE
    return out


def normal_f32(seed: int, stream: int, shape, mean: float = 0.0, std: float = 1.0, offset: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    v = normal(seed, stream, n, offset)
    return (v * std + mean).astype(np.float32).reshape(shape)
