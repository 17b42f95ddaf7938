"""Statistical quality of the counter-based dropout generator (mmnas_amd/csrc/rng.h, one MurmurHash3 finaliser
round; numpy restatement oracle/dropout_rng.py -- tests/test_kernels_gpu.py checks the two agree bit for bit on the
device).  What dropout needs from it: the right keep rate, and no visible dependence between the decisions of
neighbouring elements, of different sites of one operator call, and of consecutive calls."""
import numpy as np
import pytest

from oracle import dropout_rng as R

N = 1 << 20


def _corr(a, b):
    a = a.astype(np.float64) - a.mean()
    b = b.astype(np.float64) - b.mean()
    return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))


def _seeds(base, n):
    """The seeds mmnas_amd.ops.next_seed() hands to n consecutive operator calls."""
    return [((base * 0x9E3779B97F4A7C15) + c * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF for c in range(1, n + 1)]


@pytest.mark.parametrize('p', [0.1, 0.25, 0.5, 0.9])
def test_keep_rate(p):
    for seed in _seeds(888, 3) + [0, 1, 2**63 + 12345]:
        for site in (0, 1, 2):
            rate = R.keep_mask(seed, site, N, p).mean()
            sigma = np.sqrt(p * (1 - p) / N)
            assert abs(rate - (1 - p)) < 4.5 * sigma, (seed, site, rate)


def test_lag_autocorrelation_within_a_stream():
    """Neighbouring elements -- along a row (lag 1..3) and along a column of the tensors the masks cover (row widths
    14, 100, 256, 512, 1024, 2048 and the attention map's 100 x 100) -- are uncorrelated."""
    bound = 4.5 / np.sqrt(N)
    for seed in _seeds(888, 2):
        m = R.keep_mask(seed, 1, N + 20000, 0.1)
        for lag in (1, 2, 3, 7, 14, 64, 100, 256, 512, 1024, 2048, 10000):
            c = _corr(m[:N], m[lag:lag + N])
            assert abs(c) < bound, (seed, lag, c)


def test_sites_of_one_call_are_independent():
    bound = 4.5 / np.sqrt(N)
    for seed in _seeds(7, 2):
        ms = [R.keep_mask(seed, s, N, 0.1) for s in (0, 1, 2)]
        for i in range(3):
            for j in range(i + 1, 3):
                assert abs(_corr(ms[i], ms[j])) < bound, (seed, i, j)


def test_consecutive_calls_are_independent():
    """Seeds of consecutive operator calls differ by a fixed 64-bit increment: their streams must not be shifted or
    correlated copies of each other (same site, same indices)."""
    bound = 4.5 / np.sqrt(N)
    seeds = _seeds(888, 6)
    ms = [R.keep_mask(s, 1, N + 8, 0.1) for s in seeds]
    for i in range(len(ms) - 1):
        for lag in (0, 1, 2, 4):
            assert abs(_corr(ms[i][:N], ms[i + 1][lag:lag + N])) < bound, (i, lag)
    # seeds that differ only in the low or only in the high word
    a = R.keep_mask(0x1234567800000001, 0, N, 0.5)
    b = R.keep_mask(0x1234567800000002, 0, N, 0.5)
    c = R.keep_mask(0x1234567900000001, 0, N, 0.5)
    assert abs(_corr(a, b)) < bound and abs(_corr(a, c)) < bound


def test_threshold_bits_are_balanced():
    """The decision uses the top 24 hash bits: each of them is a fair coin over consecutive indices."""
    seed = _seeds(888, 1)[0]
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    idx = np.arange(N, dtype=np.uint64)
    h = R._fmix32(((idx * np.uint64(0x9E3779B1) + lo) & R._M32) ^ ((np.uint64(1) * np.uint64(0x85EBCA77) + hi) & R._M32))
    for bit in range(8, 32):
        f = float(((h >> np.uint64(bit)) & np.uint64(1)).mean())
        assert abs(f - 0.5) < 4.5 * 0.5 / np.sqrt(N), (bit, f)
