"""TEST INFRASTRUCTURE ONLY -- never imported by the product package (mmnas_amd/).

numpy restatement of the counter-based dropout generator that the HIP kernels use
(mmnas_amd/csrc/rng.h).  The reference (modules.py:22,135,176,256,...) uses
``nn.Dropout``, whose Philox stream cannot be reproduced outside ATen; parity with
dropout enabled is therefore checked by *mask replay*: the kernels derive every keep
decision from (seed, site, element index) with the hash below, this file derives the
same mask on the CPU, and the oracle applies it as an explicit multiplier.

    h    = fmix32((idx * 0x9E3779B1 + seed_lo) ^ (site * 0x85EBCA77 + seed_hi))
    keep = (h >> 8) >= floor(p * 2^24)
    drop(x) = keep ? x / (1 - p) : 0          (inverted dropout, modules.py semantics)

fmix32 is the MurmurHash3 32-bit finaliser (public domain, A. Appleby).
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _fmix32(h):
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def dropout_threshold(p):
    """Integer threshold on the top 24 hash bits: keep iff bits >= threshold."""
    return int(np.floor(float(p) * float(1 << 24)))


def keep_mask(seed, site, n, p):
    """Boolean keep mask for elements 0..n-1 of dropout site `site` (uint32) under `seed` (uint64)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo = np.uint64(seed & 0xFFFFFFFF)
    hi = np.uint64((seed >> 32) & 0xFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64)
    a = (idx * np.uint64(0x9E3779B1) + lo) & _M32
    b = (np.uint64(int(site) & 0xFFFFFFFF) * np.uint64(0x85EBCA77) + hi) & _M32
    h = _fmix32(a ^ b)
    return (h >> np.uint64(8)) >= np.uint64(dropout_threshold(p))


def scaled_mask(seed, site, shape, p):
    """float32 multiplier tensor (0 or 1/(1-p)) of the given shape, row-major element order."""
    n = int(np.prod(shape))
    if p <= 0.0:
        return np.ones(shape, dtype=np.float32)
    m = keep_mask(seed, site, n, p).astype(np.float32) / np.float32(1.0 - p)
    return m.reshape(shape)
