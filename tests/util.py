"""Shared helpers for the parity tests."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')

# BASELINE.json north_star: forward outputs within 1e-3 relative (fp32) of the reference
# operators; SURVEY 8(d) "Parity gate": ||a-b||_inf / ||b||_inf, same bound for gradients.
TOL = 1e-3


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    den = np.max(np.abs(b))
    num = np.max(np.abs(a - b)) if a.size else 0.0
    if den == 0:
        return float(num)
    return float(num / den)


def golden_err(npz, key, arr):
    """Relative error of `arr` against golden entry `key` (stored fully, or sample+norm)."""
    arr = np.asarray(arr, dtype=np.float32)
    if key in npz.files:
        return rel_err(arr, npz[key])
    io = key.split('|')[-1] in ('out', 'dx', 'dy')
    stride = 7 if io else 53
    samp = npz[key + '#sample']
    assert tuple(npz[key + '#shape']) == arr.shape, (key, npz[key + '#shape'], arr.shape)
    e1 = rel_err(arr.reshape(-1)[::stride], samp)
    # the stored sample's scale can be much smaller than the full tensor's; also check the norm
    n = float(np.sqrt(np.sum(arr.astype(np.float64) ** 2)))
    ref = float(npz[key + '#norm'])
    e2 = abs(n - ref) / max(ref, 1e-30)
    return max(e1, e2)


def has(npz, key):
    return key in npz.files or (key + '#sample') in npz.files
