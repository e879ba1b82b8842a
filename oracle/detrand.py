"""Deterministic, library-version-independent pseudo-random tensors for tests and goldens.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy's ``Generator`` streams and torch's Philox streams are not guaranteed stable
across library versions or devices, so goldens generated in the build container
could not be regenerated bit-for-bit on the GPU box.  Everything here is integer
splitmix64 arithmetic on uint64 arrays plus IEEE float64 scaling, which *is*
portable: the same (seed, stream, shape) always gives the same bits.
"""
import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def _mix(x):
    with np.errstate(over="ignore"):
        z = x + _M1
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        return z ^ (z >> np.uint64(31))


def _key(seed, stream):
    k = _mix(np.array([np.uint64(seed)], dtype=np.uint64))
    s = np.frombuffer(str(stream).encode(), dtype=np.uint8).astype(np.uint64)
    for b in s:
        k = _mix(k ^ b)
    return k[0]


def bits(seed, stream, n):
    """n uint64 words, a pure function of (seed, stream, index)."""
    idx = np.arange(n, dtype=np.uint64)
    return _mix(idx ^ _key(seed, stream))


def uniform(seed, stream, shape, lo=0.0, hi=1.0, dtype=np.float32):
    """U[lo,hi) with 53-bit resolution, then cast."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (bits(seed, stream, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).astype(dtype).reshape(shape)


def normal(seed, stream, shape, std=1.0, dtype=np.float32):
    """Approximately N(0,std^2): sum of 12 uniforms minus 6 (Irwin-Hall).

    Uses only +,-,* on exactly representable float64 values, so it is bit-stable on
    every platform (Box-Muller would depend on libm's log/cos in the last ulp).
    """
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float64)
    for k in range(12):
        acc += (bits(seed, f"{stream}/ih{k}", n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return ((acc - 6.0) * std).astype(dtype).reshape(shape)


def integers(seed, stream, shape, lo, hi):
    """Integers in [lo,hi) (modulo bias is irrelevant for tests)."""
    n = int(np.prod(shape)) if len(shape) else 1
    r = bits(seed, stream, n) % np.uint64(hi - lo)
    return (r.astype(np.int64) + lo).reshape(shape)


def permutation(seed, stream, n):
    """A permutation of range(n): argsort of distinct 64-bit keys."""
    return np.argsort(bits(seed, stream, n), kind="stable").astype(np.int64)
