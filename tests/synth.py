"""Deterministic, platform-independent synthetic SAE weights / activations for the parity tests.

Counter-based: value(seed, i) depends only on (seed, i), is produced with integer arithmetic plus a
single f32 multiply (no libm transcendental), so the golden-fixture generator (run once in the
build container, next to the reference) and the tests (run anywhere) regenerate bit-identical
arrays without shipping gigabytes of weights.

Distribution: Irwin-Hall(4) of 16-bit uniforms, centred, scaled to unit variance: bell-shaped,
bounded at +-3.46 sigma.  Activations additionally get a few "massive" outlier dimensions like a
real LLM residual stream.
"""
from __future__ import annotations

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_IH4_SIGMA = np.float32(37837.22)  # sqrt(4 * (65536^2 - 1) / 12)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def normalish(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n unit-variance f32 samples for counters offset .. offset+n-1 of stream `seed`."""
    with np.errstate(over="ignore"):
        ctr = np.arange(offset, offset + n, dtype=np.uint64)
        z = _splitmix64((ctr + np.uint64(1)) * _GOLD + np.uint64(seed) * _M2)
    s = ((z & np.uint64(0xFFFF)) + ((z >> np.uint64(16)) & np.uint64(0xFFFF))
         + ((z >> np.uint64(32)) & np.uint64(0xFFFF)) + (z >> np.uint64(48))).astype(np.int64)
    s -= 2 * 65535
    return s.astype(np.float32) / _IH4_SIGMA


def bf16_round(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even f32 -> bf16 -> f32 (values representable in bf16)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xFFFF0000)
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def sae_weights(d: int, N: int, seed: int = 1, chunk: int = 1 << 22):
    """(W_enc[N,d], b_enc[N], W_dec[N,d], b_dec[d]) f32.  Rows ~unit norm (entries ~N(0,1/d))."""
    scale = np.float32(1.0 / np.sqrt(d))

    def mat(s):
        out = np.empty(N * d, dtype=np.float32)
        for o in range(0, N * d, chunk):
            m = min(chunk, N * d - o)
            out[o:o + m] = normalish(s, m, o) * scale
        return out.reshape(N, d)

    W_enc = mat(seed * 4 + 0)
    W_dec = mat(seed * 4 + 1)
    b_enc = normalish(seed * 4 + 2, N) * np.float32(0.05)
    b_dec = normalish(seed * 4 + 3, d) * np.float32(0.1)
    return W_enc, b_enc, W_dec, b_dec


def activations(T: int, d: int, seed: int = 0, bf16: bool = True, n_outlier: int = 4):
    """x[T,d]: unit-variance residual-stream stand-in with a few x20 outlier dims; bf16-valued
    by default (the LLM hands the hook bf16/fp16 tensors; sae.py:174 up-casts them)."""
    x = normalish(1000 + seed, T * d).reshape(T, d)
    mu = normalish(2000 + seed, d) * np.float32(0.25)
    x = x + mu
    for j in range(n_outlier):
        x[:, (j * 977 + 13) % d] *= np.float32(20.0)
    return bf16_round(x) if bf16 else x
