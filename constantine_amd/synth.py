"""Synthetic benchmark inputs: seeded scalars (the points come from the device generator, ctt_hip_gen_points).

Mirrors the distribution of the reference's bench inputs (benchmarks/bench_elliptic_parallel_template.nim:78-102,
helpers/prng_unsafe.nim:166-183): scalars uniform in [0, 2^bits), NOT reduced modulo the group order.  The generator is
splitmix64 over (seed, index) so that any slice of the sequence can be produced on its own (every rank of a sharded run
makes only its pairs).  oracle/cref.py carries its own copy of the same definition for the tests.
"""
import numpy as np


def synth_scalars(seed: int, n: int, bits: int, first: int = 0) -> np.ndarray:
    """(n, 32) uint8, little-endian 256-bit integers below 2^bits; element i depends only on (seed, first + i)."""
    idx = (np.arange(first, first + n, dtype=np.uint64)[:, None] * np.uint64(4) + np.arange(4, dtype=np.uint64)[None, :])
    with np.errstate(over="ignore"):
        x = idx + np.uint64(seed & (2**64 - 1))
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    top = bits - 192
    z[:, 3] &= np.uint64((1 << top) - 1)
    return z.view(np.uint8).reshape(n, 32).copy()
