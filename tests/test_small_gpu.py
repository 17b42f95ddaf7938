"""Short-sequence operators (small.hip): the one-launch forms that the language stream's operators take (S <= 16, heads
of 64, d in {256, 512}) against the fp64 oracle and against the general multi-launch path on the same inputs."""
import numpy as np
import pytest
import torch

from oracle import mmnas_oracle as O   # noqa: F401  (oracle_runner needs the package importable)
from tests import oracle_runner as R
from tests.golden import cases
from tests.test_ops_gpu import _drop_sites, run_hip_op
from tests.util import TOL, rel_err

pytestmark = pytest.mark.gpu


def _check(got, ref, tol=TOL, floor=1e-4):
    # (a gradient that is mathematically zero -- dWq / dWk with a single key: the softmax is constant -- is rounding
    #  noise on both sides: it is measured against the operator's gradient scale)
    gscale = max(np.abs(ref[k]).max() for k in ref if k != 'out')
    for k in ref:
        den = max(np.abs(ref[k]).max(), floor * (gscale if k != 'out' else 1.0))
        e = float(np.abs(got[k].astype(np.float64) - ref[k]).max() / den)
        assert e <= tol, (k, e)


@pytest.fixture
def small_ops():
    from mmnas_amd import _lib as L
    lib = L.lib()
    prev = lib.mmnas_set_small_ops(1)
    prev_b = lib.mmnas_set_small_bwd(1)
    prev_f = lib.mmnas_set_small_ffn(1)      # (opt-in in the product: measured neutral; the kernel is kept correct here)
    yield lib
    lib.mmnas_set_small_ops(prev)
    lib.mmnas_set_small_bwd(prev_b)
    lib.mmnas_set_small_ffn(prev_f)


@pytest.mark.parametrize('name', ['self_att_64'])
@pytest.mark.parametrize('dims', [dict(B=64, Sx=14, Sy=3, HSIZE=256), dict(B=5, Sx=16, Sy=3, HSIZE=512),
                                  dict(B=7, Sx=5, Sy=3, HSIZE=256), dict(B=2, Sx=1, Sy=3, HSIZE=512)])
@pytest.mark.parametrize('nr', [(True, True), (False, False), (True, False)])
def test_short_sequence_op_vs_oracle(name, dims, nr, small_ops):
    case = cases.op_case(name, nr[0], nr[1], 99 + dims['Sx'], dims)
    got = run_hip_op(case)
    ref = R.run_oracle_op(case, dtype=torch.float64)
    _check(got, ref, floor=1e-2 if dims['Sx'] == 1 else 1e-4)
    # the one-launch forward followed by the GENERAL backward (either backward follows either forward)
    prev_b = small_ops.mmnas_set_small_bwd(0)
    try:
        mixed = run_hip_op(case)
    finally:
        small_ops.mmnas_set_small_bwd(prev_b)
    _check(mixed, ref, floor=1e-2 if dims['Sx'] == 1 else 1e-4)
    small_ops.mmnas_set_small_ops(0)
    gen = run_hip_op(case)
    gscale = max(np.abs(gen[k]).max() for k in gen if k != 'out')
    for other in (gen, mixed):
        for k in got:          # same arithmetic up to summation order
            den = max(np.abs(other[k]).max(), 1e-2 * (gscale if k != 'out' else 1.0))
            assert np.abs(got[k] - other[k]).max() / den <= (1e-4 if dims['Sx'] == 1 else 2e-5), k


@pytest.mark.parametrize('name', ['self_att_64'])
@pytest.mark.parametrize('dims', [dict(B=9, Sx=14, Sy=3, HSIZE=256), dict(B=3, Sx=16, Sy=3, HSIZE=512), dict(B=64, Sx=14, Sy=3, HSIZE=256)])
def test_short_sequence_op_dropout_replay(name, dims, small_ops, monkeypatch):
    from mmnas_amd import ops
    seed, p = 0x1234ABCD0F0F0F0F, 0.1
    monkeypatch.setattr(ops, 'next_seed', lambda: seed)
    case = cases.op_case(name, True, True, 4242, dims)
    got = run_hip_op(case, train=True, drop_p=p)
    case['cfg'].DROPOUT_R = 0.0
    ref = R.run_oracle_op(case, drops=_drop_sites(case, seed, p), dtype=torch.float64)
    _check(got, ref)
    ref0 = R.run_oracle_op(case, dtype=torch.float64)
    assert rel_err(got['out'], ref0['out']) > 1e-2
    prev_b = small_ops.mmnas_set_small_bwd(0)      # the general backward replays the same masks
    try:
        mixed = run_hip_op(case, train=True, drop_p=p)
    finally:
        small_ops.mmnas_set_small_bwd(prev_b)
    _check(mixed, ref)
    for k in got:
        assert np.abs(got[k] - mixed[k]).max() <= 2e-5 * max(np.abs(mixed[k]).max(), 1e-3), k


def test_fully_padded_sample_is_uniform_attention(small_ops):
    """masked_fill(-1e9) semantics (modules.py:194): a sample whose keys are all padding attends uniformly."""
    case = cases.op_case('self_att_64', True, True, 5, dict(B=4, Sx=14, Sy=3, HSIZE=256))
    case['x_mask'][1, ...] = True
    got = run_hip_op(case)
    ref = R.run_oracle_op(case, dtype=torch.float64)
    _check(got, ref)


# ---- FeedForward forward in one launch (ffn_small_fwd_kernel: d = 256, hidden 1024, <= 1024 rows in groups of 16) ----
@pytest.mark.parametrize('dims', [dict(B=64, Sx=14, Sy=3, HSIZE=256), dict(B=7, Sx=5, Sy=3, HSIZE=256), dict(B=3, Sx=16, Sy=3, HSIZE=256),
                                  dict(B=1, Sx=1, Sy=3, HSIZE=256)])
@pytest.mark.parametrize('nr', [(True, True), (False, False), (True, False)])
@pytest.mark.parametrize('mode', [1, 2])       # 4 hidden slices of 256 / 8 of 128 (two workgroups per CU)
def test_short_feed_forward_vs_oracle_and_general_path(dims, nr, mode, small_ops):
    small_ops.mmnas_set_small_ffn(mode)
    case = cases.op_case('feed_forward', nr[0], nr[1], 77 + dims['Sx'], dims)
    got = run_hip_op(case)
    ref = R.run_oracle_op(case, dtype=torch.float64)
    _check(got, ref)
    small_ops.mmnas_set_small_ops(0)
    gen = run_hip_op(case)
    gscale = max(np.abs(gen[k]).max() for k in gen if k != 'out')
    for k in got:          # same arithmetic up to summation order
        den = max(np.abs(gen[k]).max(), 1e-2 * (gscale if k != 'out' else 1.0))
        assert np.abs(got[k] - gen[k]).max() / den <= 2e-5, k


@pytest.mark.parametrize('dims', [dict(B=9, Sx=14, Sy=3, HSIZE=256), dict(B=64, Sx=14, Sy=3, HSIZE=256)])
@pytest.mark.parametrize('mode', [1, 2])
def test_short_feed_forward_dropout_replay(dims, mode, small_ops, monkeypatch):
    from mmnas_amd import ops
    small_ops.mmnas_set_small_ffn(mode)
    seed, p = 0x0BADC0DE12345678, 0.1
    monkeypatch.setattr(ops, 'next_seed', lambda: seed)
    case = cases.op_case('feed_forward', True, True, 4343, dims)
    got = run_hip_op(case, train=True, drop_p=p)
    case['cfg'].DROPOUT_R = 0.0
    ref = R.run_oracle_op(case, drops=_drop_sites(case, seed, p), dtype=torch.float64)
    _check(got, ref)
    small_ops.mmnas_set_small_ops(0)        # the general path draws the same masks
    gen = run_hip_op(case, train=True, drop_p=p)
    for k in got:
        assert np.abs(got[k] - gen[k]).max() <= 2e-5 * max(np.abs(gen[k]).max(), 1e-3), k


def test_short_operators_really_take_the_one_launch_kernels(small_ops):
    """(a silent fall-back to the general path would make the tests above vacuous): the launch counts of the library's
    'small_ops' kernel class -- FeedForward: 1 forward launch; SelfAtt: 1 forward + 1 backward"""
    import ctypes as C
    from mmnas_amd import _lib as L
    lib = small_ops
    si = L.K_NAMES.index('small_ops')

    def launches(name):
        case = cases.op_case(name, True, True, 3, dict(B=64, Sx=14, Sy=3, HSIZE=256))
        arr = (L.ProfStat * len(L.K_NAMES))()
        L.check(lib.mmnas_prof_enable(1))
        try:
            run_hip_op(case)
            torch.cuda.synchronize()
            L.check(lib.mmnas_prof_collect(arr))
        finally:
            L.check(lib.mmnas_prof_enable(0))
        return arr[si].launches

    assert launches('feed_forward') == 1
    assert launches('self_att_64') == 2
    lib.mmnas_set_small_ops(0)
    assert launches('feed_forward') == 0 and launches('self_att_64') == 0
