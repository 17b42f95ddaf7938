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


def esample(a, n=64):
    """The strided element sample tests/golden/make_golden.py::esample stores (<= n elements, fixed stride)."""
    flat = np.ascontiguousarray(np.asarray(a), dtype=np.float32).reshape(-1)
    return flat[::max(1, flat.size // n)][:n]


def is_rel_path(key):
    """Parameters whose gradient is a relation-bias contraction (linear_r of a RelSelfAtt, the shared stem linear_{x,y}_rel)."""
    return 'linear_r.' in key or 'linear_y_rel.' in key or 'linear_x_rel.' in key


# Two fp32 evaluations of the SAME relation-path gradient through different kernels (per-operator relfused.hip on the
# vector pipe vs all operators at once on the MFMA, relmulti.hip; or packed vs padded rows) add 1/r-weighted terms of random
# sign in different orders: measured 2e-4 ... 3e-4 of the tensor's largest entry apart (MI355X, round 5), both within the
# float64-yardstick bound of check_grad_samples.  Self-consistency tests use this for those keys and their own (tight)
# bound for every other parameter.
REL_PATH_SELF_TOL = float(os.environ.get('MMNAS_TEST_REL_SELF_TOL', '5e-4'))   # (env: the sweep that set it)

KAPPA = 3e-4   # relation-path gradients: bound on |error| / (sum of the absolute values of the entry's terms)


# what check_grad_samples measured in this process: tag -> worst relative error (non-relation keys), worst error / bound;
# tests/conftest.py writes it to gpurun_out/grad_check_stats.json at the end of a session (the numbers DESIGN 2 quotes)
GRAD_STATS = {}


def check_grad_samples(npz, tag, grads, tol=1e-3, skip=lambda k: False, kappa=None):
    """Element-wise check of EVERY parameter gradient of a network against the reference's strided samples
    (`tag + 'gs_keys' / 'gs' / 'gs_off'`): a sign or permutation error inside a weight gradient keeps its norm, not these.
    grads: name -> array (None / missing = no gradient).  Tolerance: `tol` = 1e-3 (SURVEY 8(d)'s bar; round 6: EVERY key is
    judged against the reference's FLOAT64 run -- `gs64*` now holds all parameters -- where rounds 3-5 judged most keys against
    the reference's own fp32 run and needed 3e-3) of the tensor's largest sampled reference element, floored at 1e-3 of the
    largest over all tensors (round-off-sized gradients).

    The relation-path gradients (`linear_r`, `linear_{x,y}_rel`) are 1/r-weighted sums of random sign behind
    log(clamp(relu(.))) that cancel to a small fraction of their terms.  For them the yardstick is the reference run in
    FLOAT64 (`gs64*`, make_golden.grad_samples64), and the bound is the ordinary `tol` on the entry OR KAPPA on the entry's
    cancellation scale (`gs64_scale`: the sum of the absolute values of its terms, from the same float64 run), whichever is
    larger -- not a blanket looser tolerance.  Measured on MI355X (tools/tmp-style probe, nets.npz
    full|vqa|mmnas_vqa, the worst entry: dag.15 linear_r.bias = 1.5e-3 against a scale of 4.9e-2): 7e-6 ... 1.2e-4 of the
    scale depending on which LSTM / relation kernels ran upstream (the same entry moves by 3.5e-3 of its value when only
    the LSTM kernel changes: conditioning, not a kernel property); the reference's own fp32 run: <= 1e-5 of the scale."""
    keys = [str(k) for k in npz[tag + 'gs_keys']]
    off = npz[tag + 'gs_off']
    gs = npz[tag + 'gs']
    ref64, scale64 = {}, {}
    if tag + 'gs64_keys' in npz.files:
        o64, g64 = npz[tag + 'gs64_off'], npz[tag + 'gs64']
        sc = npz[tag + 'gs64_scale'] if tag + 'gs64_scale' in npz.files else None
        for i, k in enumerate(npz[tag + 'gs64_keys']):
            ref64[str(k)] = g64[o64[i]:o64[i + 1]]
            if sc is not None:
                scale64[str(k)] = sc[o64[i]:o64[i + 1]]
    top = float(np.max(np.abs(gs))) if gs.size else 0.0
    checked = 0
    st = GRAD_STATS.setdefault(tag, {'worst_rel': 0.0, 'worst_key': None, 'worst_of_bound': 0.0, 'worst_of_bound_key': None, 'fp64_keys': 0, 'keys': 0})
    for i, k in enumerate(keys):
        if skip(k):
            continue
        ref = ref64.get(k, gs[off[i]:off[i + 1]])
        g = grads.get(k)
        assert g is not None, ('no gradient for', k)
        mine = esample(g)
        assert mine.shape == ref.shape, (k, mine.shape, ref.shape)
        err = np.abs(mine.astype(np.float64) - ref)
        bound = tol * max(float(np.max(np.abs(ref))), 1e-3 * top) * np.ones_like(err)
        if k in scale64:
            bound = np.maximum(bound, (KAPPA if kappa is None else kappa) * scale64[k])
        if k in ref64:
            # An element where the reference's OWN fp32 run is more than half the bound away from its float64 run has no valid
            # float64 yardstick: a branch upstream (ReLU, the 1e-6 clamp, a masked maximum) fell on different sides in the two
            # precisions.  Both runs are the reference; such an element is judged against the fp32 run at the same bound.  (Round
            # 6, the arch step at B = 64: ONE sample of embedding.weight, 8.5e-3 of 3.8 apart in the reference itself -- and the
            # GPU result within 1e-5 of the reference's fp32 value.)
            ref32 = gs[off[i]:off[i + 1]].astype(np.float64)
            flipped = np.abs(ref32 - ref) > 0.5 * bound
            if flipped.any():
                err = np.where(flipped, np.abs(mine.astype(np.float64) - ref32), err)
                st['fp32_yardstick_elements'] = st.get('fp32_yardstick_elements', 0) + int(flipped.sum())
        bad = err > bound
        assert not bad.any(), (k, float(err[bad].max()), float(np.max(np.abs(ref))), 'fp64 yardstick' if k in ref64 else 'fp32',
                               float(scale64[k][bad].max()) if k in scale64 else None)
        checked += 1
        rel = float(err.max()) / max(float(np.max(np.abs(ref))), 1e-3 * top, 1e-30)      # the error the bound is written on
        frac = float(np.max(err / bound))
        st['keys'] += 1
        st['fp64_keys'] += int(k in ref64)
        if rel > st['worst_rel'] and not (k in scale64 and float(np.max(scale64[k])) > 0.0):
            st['worst_rel'], st['worst_key'] = rel, k
        if frac > st['worst_of_bound']:
            st['worst_of_bound'], st['worst_of_bound_key'] = frac, k
    return checked
