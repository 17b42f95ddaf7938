"""Kernel-level parity (GPU): each C-ABI kernel against a plain torch fp64 CPU reference of the
same op.  Tolerance 1e-3 relative (BASELINE.json north_star); fp32-MFMA kernels land near 1e-6."""
import math

import numpy as np
import pytest
import torch

from tests.util import TOL, rel_err

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def _lib():
    from mmnas_amd import _lib as L
    return L


def g(t):
    return torch.as_tensor(t).to(DEV).contiguous()


def rnd(rs, *shape):
    return rs.standard_normal(shape).astype(np.float32)


# ----------------------------------------------------------------------------- dropout RNG
def test_dropout_mask_matches_numpy_restatement():
    from mmnas_amd import ops
    from oracle import dropout_rng
    for p, seed, site, n in ((0.1, 0x123456789ABCDEF0, 0, 100003), (0.5, 7, 1, 4096), (0.0, 9, 2, 100)):
        got = ops.dropout_mask(n, p, seed, site, DEV).cpu().numpy()
        ref = dropout_rng.scaled_mask(seed, site, (n,), p)
        assert np.array_equal(got, ref)
        if p > 0:
            assert abs((got == 0).mean() - p) < 0.01


# ----------------------------------------------------------------------------- GEMM
@pytest.fixture
def gemm_tuning():
    """Set MMNAS_GEMM_* scheduling knobs for one test (the library caches them: reload after every change)."""
    import os
    import mmnas_amd._lib as L
    saved = {}

    def set_knobs(**kw):
        for k, v in kw.items():
            name = 'MMNAS_GEMM_' + k.upper()
            saved.setdefault(name, os.environ.get(name))
            if v is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = str(v)
        L.lib().mmnas_gemm_reload_tuning()

    yield set_knobs
    for name, v in saved.items():
        if v is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = v
    L.lib().mmnas_gemm_reload_tuning()


@pytest.mark.parametrize('layout', ['NT', 'NN', 'TN'])
@pytest.mark.parametrize('M,N,K', [(896, 512, 512), (300, 192, 160), (64, 64, 32), (21, 3129, 1024),
                                   (130, 1, 64), (257, 130, 100), (6400, 256, 256)])
@pytest.mark.parametrize('tile', [0, 64, 128])
@pytest.mark.parametrize('mode', [6, 0])     # the default bf16x6 products and the fp32-MFMA kernels
def test_gemm_layouts(layout, M, N, K, tile, mode, gemm_tuning):
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    if tile and (M * N > 400000):
        pytest.skip('big case only with the default tile choice')
    rs = np.random.RandomState(M * 7 + N * 3 + K)
    if layout == 'NT':
        A, B = rnd(rs, M, K), rnd(rs, N, K)
        ref = torch.from_numpy(A).double() @ torch.from_numpy(B).double().t()
        lda, ldb = K, K
    elif layout == 'NN':
        A, B = rnd(rs, M, K), rnd(rs, K, N)
        ref = torch.from_numpy(A).double() @ torch.from_numpy(B).double()
        lda, ldb = K, N
    else:
        A, B = rnd(rs, K, M), rnd(rs, K, N)
        ref = torch.from_numpy(A).double().t() @ torch.from_numpy(B).double()
        lda, ldb = M, N
    Ad, Bd = g(A), g(B)
    Cd = torch.zeros(M, N, device=DEV)
    gemm_tuning(tile=tile or None, split=mode)
    split = 1
    if layout == 'TN' and K >= 256:
        split = 3
    lay = {'NT': L.GEMM_NT, 'NN': L.GEMM_NN, 'TN': L.GEMM_TN}[layout]
    ops.gemm(lay, [dict(M=M, A=[Ad], B=[Bd], C=Cd)], N, K, lda, ldb, N, split_k=split)
    err = rel_err(Cd.cpu().numpy(), ref.numpy())
    assert err < 1e-5, err


@pytest.mark.parametrize('layout', ['NT', 'NN', 'TN'])
@pytest.mark.parametrize('M,N,K', [(896, 512, 512), (300, 192, 160), (64, 64, 32), (257, 132, 96), (6400, 256, 256)])
@pytest.mark.parametrize('mode,tol', [(3, 2e-5), (6, 3e-6), (1, 6e-3)])
@pytest.mark.parametrize('tile', [0, 128])
def test_gemm_bf16_split_modes(layout, M, N, K, mode, tol, tile, gemm_tuning):
    """MMNAS_GEMM_SPLIT modes (6 = the default): operands split into 2 / 3 bf16 parts, 3 / 6 bf16-MFMA products, fp32 accumulate.
    Bounds (max-norm relative, against fp64): 2^-16-class for bf16x3, fp32-class for bf16x6 -- both far inside the
    1e-3 parity tolerance.  Mode 1 (ONE product of bf16-rounded operands: the reduced-precision flavour of BASELINE
    configs[4], never a default) keeps 8 mantissa bits per operand: 2^-8-class, bounded here at 6e-3 and -- against the
    product of the ROUNDED operands in fp64, which is what it computes -- at fp32 accumulation error.  Operand scales spread over 2^+-20 to exercise the exponent range the parts share with fp32."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    if tile and M * N > 400000:
        pytest.skip('big case only with the default tile choice')
    rs = np.random.RandomState(M + N + K + mode)
    sa, sb = np.float32(2.0 ** rs.randint(-20, 20)), np.float32(2.0 ** rs.randint(-20, 20))
    td = lambda a: torch.from_numpy(a).double()
    if layout == 'NT':
        A, B = rnd(rs, M, K) * sa, rnd(rs, N, K) * sb
        ref, lda, ldb = td(A) @ td(B).t(), K, K
    elif layout == 'NN':
        A, B = rnd(rs, M, K) * sa, rnd(rs, K, N) * sb
        ref, lda, ldb = td(A) @ td(B), K, N
    else:
        A, B = rnd(rs, K, M) * sa, rnd(rs, K, N) * sb
        ref, lda, ldb = td(A).t() @ td(B), M, N
    C = torch.zeros(M, N, device=DEV)
    gemm_tuning(split=mode, tile=tile or None)
    lay = {'NT': L.GEMM_NT, 'NN': L.GEMM_NN, 'TN': L.GEMM_TN}[layout]
    ops.gemm(lay, [dict(M=M, A=[g(A)], B=[g(B)], C=C)], N, K, lda, ldb, N, accumulate=(layout == 'TN'))
    err = rel_err(C.cpu().numpy(), ref.numpy())
    assert err < tol, err
    if mode == 1 and err > 1e-5:    # (shapes outside the buffer-load path stay on the fp32 MFMA whatever the mode)
        rb = lambda a: torch.from_numpy(a).bfloat16().double()
        ref1 = {'NT': lambda: rb(A) @ rb(B).t(), 'NN': lambda: rb(A) @ rb(B), 'TN': lambda: rb(A).t() @ rb(B)}[layout]()
        assert rel_err(C.cpu().numpy(), ref1.numpy()) < 3e-6


@pytest.mark.parametrize('layout,M,N,K', [('NT', 6400, 256, 256), ('NT', 6400, 1024, 256), ('NN', 6400, 256, 1024),
                                          ('TN', 256, 1024, 6400), ('NT', 6400, 2048, 512), ('NN', 6400, 512, 2048),
                                          ('TN', 2048, 512, 6400), ('NT', 896, 512, 512)])
def test_default_products_are_fp32_grade(layout, M, N, K, gemm_tuning):
    """The default GEMM path (6 bf16-MFMA products of exactly split operands, dropped terms < 2^-24 |a||b|) against the
    fp32-MFMA path on the workloads' shapes, both measured against fp64: the RMS error is not larger (15 % slack for the
    different summation order; 1.5x on the max-norm, a noisy statistic of a million rounding errors)."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    rs = np.random.RandomState(M + 3 * N + 7 * K)
    td = lambda a: torch.from_numpy(a).double()
    if layout == 'NT':
        A, B = rnd(rs, M, K), rnd(rs, N, K)
        ref, lda, ldb = td(A) @ td(B).t(), K, K
    elif layout == 'NN':
        A, B = rnd(rs, M, K), rnd(rs, K, N)
        ref, lda, ldb = td(A) @ td(B), K, N
    else:
        A, B = rnd(rs, K, M), rnd(rs, K, N)
        ref, lda, ldb = td(A).t() @ td(B), M, N
    lay = {'NT': L.GEMM_NT, 'NN': L.GEMM_NN, 'TN': L.GEMM_TN}[layout]
    err = {}
    for mode in (0, 6):
        gemm_tuning(split=mode)
        C = torch.zeros(M, N, device=DEV)
        ops.gemm(lay, [dict(M=M, A=[g(A)], B=[g(B)], C=C)], N, K, lda, ldb, N, accumulate=(layout == 'TN'))
        d = C.cpu().double() - ref
        err[mode] = (float(d.abs().max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
    assert err[6][0] <= 1.5 * err[0][0] and err[6][1] <= 1.15 * err[0][1], err
    assert err[6][0] < 3e-6, err


def _both_products(A, B, gemm_tuning):
    """A [M,K] x B[N,K]^T on the default (bf16x6) path and on the fp32 MFMA, with the fp64 product."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    (M, K), N = A.shape, B.shape[0]
    ref = torch.from_numpy(A).double() @ torch.from_numpy(B).double().t()
    out = {}
    for mode in (0, 6):
        gemm_tuning(split=mode)
        C = torch.zeros(M, N, device=DEV)
        ops.gemm(L.GEMM_NT, [dict(M=M, A=[g(A)], B=[g(B)], C=C)], N, K, K, K, N)
        out[mode] = C.cpu().double()
    return ref, out


@pytest.mark.parametrize('sa,sb', [(1e-30, 1e20), (1e30, 1e-20), (1e-30, 1e30), (1e-30, 1e-5), (1e18, 1e18)])
def test_default_products_operand_range(sa, sb, gemm_tuning):
    """Operand magnitudes across the fp32 range: bf16 has fp32's exponent, so the three parts of an operand are exact down
    to |x| ~ 1e-35 and up to the overflow threshold; the error against fp64 stays at the fp32 MFMA's."""
    rs = np.random.RandomState(17)
    A, B = (rnd(rs, 256, 512) * np.float32(sa)), (rnd(rs, 192, 512) * np.float32(sb))
    ref, out = _both_products(A, B, gemm_tuning)
    e = {m: float((out[m] - ref).abs().max() / ref.abs().max()) for m in out}
    rms = {m: float((out[m] - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) for m in out}
    # (the max-norm of ~50 k rounding errors is itself a noisy statistic: 1.5x slack there, 1.15x on the RMS)
    assert np.isfinite(e[6]) and e[6] <= 1.5 * e[0] and rms[6] <= 1.15 * rms[0] and e[6] < 3e-6, (e, rms)


def test_default_products_mixed_row_scales(gemm_tuning):
    """Rows of very different magnitude inside one operand (e^-27 .. e^+27 per row): every output element is accurate
    relative to its OWN row / column scale, not only relative to the largest element of the product."""
    rs = np.random.RandomState(18)
    A = (rnd(rs, 256, 512) * np.exp(rs.uniform(-27, 27, (256, 1)))).astype(np.float32)
    B = (rnd(rs, 192, 512) * np.exp(rs.uniform(-27, 27, (192, 1)))).astype(np.float32)
    ref, out = _both_products(A, B, gemm_tuning)
    scale = torch.from_numpy(np.sqrt((A.astype(np.float64) ** 2).sum(1))[:, None] * np.sqrt((B.astype(np.float64) ** 2).sum(1))[None, :])
    e = {m: float(((out[m] - ref).abs() / scale).max()) for m in out}
    assert e[6] <= 1.2 * e[0] + 1e-9 and e[6] < 1e-6, e


def test_default_products_where_the_residuals_underflow(gemm_tuning):
    """The documented limit of the split: below |x| ~ 1e-35 the second and third parts of an operand (2^-8, 2^-16 of it)
    fall under bf16's smallest normal number and are flushed, so an operand that is tiny THROUGHOUT is multiplied with
    fewer bits (measured 2e-4 at 1e-37; the fp32 MFMA keeps 5e-7).  It stays inside the 1e-3 parity gate, and an operand
    with ordinary entries beside tiny ones is unaffected (the test above).  MMNAS_GEMM_SPLIT=0 selects the fp32 MFMA."""
    rs = np.random.RandomState(19)
    A, B = (rnd(rs, 256, 512) * np.float32(1e-37)), (rnd(rs, 192, 512) * np.float32(1e30))
    ref, out = _both_products(A, B, gemm_tuning)
    e = {m: float((out[m] - ref).abs().max() / ref.abs().max()) for m in out}
    assert e[0] < 3e-6 and e[6] < 1e-3, e


@pytest.mark.parametrize('mode', [3, 6])
def test_gemm_bf16_split_stream_k_and_epilogue(mode, gemm_tuning):
    """Split modes under the stream-K schedule with grouped problems, K segments and the full epilogue."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    gemm_tuning(split=mode, sk=2, min_units=1, wgs=33)
    rs = np.random.RandomState(mode)
    td = lambda a: torch.from_numpy(a).double()
    Ms, N, K = [300, 77, 130], 200, 160
    A = [[rnd(rs, m, K) for _ in range(2)] for m in Ms]
    B = [[rnd(rs, N, K) for _ in range(2)] for _ in Ms]
    bias = [rnd(rs, N) for _ in Ms]
    res = [rnd(rs, m, N) for m in Ms]
    Cs = [torch.full((m, N), float('nan'), device=DEV) for m in Ms]
    groups = [dict(M=m, A=[g(x) for x in a], B=[g(x) for x in b], C=c, bias=g(bi), residual=g(r))
              for m, a, b, c, bi, r in zip(Ms, A, B, Cs, bias, res)]
    ops.gemm(L.GEMM_NT, groups, N, K, K, K, N, nseg=2, ldres=N, relu=True)
    for i in range(len(Ms)):
        ref = torch.relu(sum(td(a) @ td(b).t() for a, b in zip(A[i], B[i])) + td(bias[i])) + td(res[i])
        assert rel_err(Cs[i].cpu().numpy(), ref.numpy()) < (2e-5 if mode == 3 else 3e-6)


@pytest.mark.parametrize('M,nin,nout', [(6400, 512, 512), (896, 256, 1024), (300, 132, 68), (6400, 2048, 512), (64, 32, 32)])
@pytest.mark.parametrize('pair', [1, 0])
@pytest.mark.parametrize('split', [0, 6])
def test_gemm_pair_matches_two_launches(M, nin, nout, pair, split, gemm_tuning):
    """mmnas_gemm_pair (data gradient NN + weight gradient TN of one linear layer in one launch, the second section
    behind the first): grouped weight gradients, K segments, residual and gate epilogues on the data gradient;
    odd shapes fall back to two launches inside the call.  Same results as the fp64 products either way."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    gemm_tuning(pair=pair, split=split)
    rs = np.random.RandomState(M + nin + nout)
    td = lambda a: torch.from_numpy(a).double()
    dy, x, W, res = rnd(rs, M, nout), rnd(rs, M, nin), rnd(rs, nout, nin), rnd(rs, M, nin)
    gate = (rnd(rs, M, nin) > 0).astype(np.float32)
    dW0 = rnd(rs, nout, nin)
    for use_gate in (False, True):
        dx = torch.full((M, nin), float('nan'), device=DEV)
        dW = g(dW0.copy())
        grp = dict(M=M, A=[g(dy)], B=[g(W)], C=dx)
        kw = {}
        if use_gate:
            grp['gate'] = g(gate)
            kw = dict(ldgate=nin, gate_scale=1.25)
        else:
            grp['residual'] = g(res)
            kw = dict(ldres=nin)
        dg = ops.gemm_desc(L.GEMM_NN, [grp], nin, nout, nout, nin, nin, **kw)
        wg = ops.gemm_desc(L.GEMM_TN, [dict(M=nout, A=[g(dy)], B=[g(x)], C=dW)], nin, M, nout, nin, nin, accumulate=True)
        ops.gemm_pair(dg, wg)
        ref_dx = td(dy) @ td(W)
        ref_dx = torch.where(td(gate) > 0, ref_dx * 1.25, torch.zeros_like(ref_dx)) if use_gate else ref_dx + td(res)
        ref_dW = td(dW0) + td(dy).t() @ td(x)
        assert rel_err(dx.cpu().numpy(), ref_dx.numpy()) < 1e-5
        assert rel_err(dW.cpu().numpy(), ref_dW.numpy()) < 1e-5
    # grouped weight gradients (q, k, v) + a 3-segment data gradient, as the self-attention backward issues them
    d3 = [rnd(rs, M, nout) for _ in range(3)]
    W3 = [rnd(rs, nout, nin) for _ in range(3)]
    dWs = [torch.zeros(nout, nin, device=DEV) for _ in range(3)]
    dx = torch.empty(M, nin, device=DEV)
    d3d, W3d, xd, resd = [g(a) for a in d3], [g(b) for b in W3], g(x), g(res)   # (descriptors hold raw pointers)
    dg = ops.gemm_desc(L.GEMM_NN, [dict(M=M, A=d3d, B=W3d, C=dx, residual=resd)], nin, nout, nout, nin, nin, nseg=3, ldres=nin)
    wg = ops.gemm_desc(L.GEMM_TN, [dict(M=nout, A=[a], B=[xd], C=c) for a, c in zip(d3d, dWs)], nin, M, nout, nin, nin,
                       accumulate=True)
    ops.gemm_pair(dg, wg)
    ref = sum(td(a) @ td(b) for a, b in zip(d3, W3)) + td(res)
    assert rel_err(dx.cpu().numpy(), ref.numpy()) < 1e-5
    for a, c in zip(d3, dWs):
        assert rel_err(c.cpu().numpy(), (td(a).t() @ td(x)).numpy()) < 1e-5


@pytest.mark.parametrize('M,N,K', [(6400, 2048, 512), (300, 132, 96), (896, 512, 512), (64, 64, 32)])
@pytest.mark.parametrize('sk', [1, 2])
def test_gemm_colsum_epilogue(M, N, K, sk, gemm_tuning):
    """colsum[n] += sum_m C[m, n] of the stored values (gate epilogue applied), also on tiles cut by stream-K."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    gemm_tuning(sk=sk, min_units=1 if sk == 2 else None)
    rs = np.random.RandomState(M + N)
    td = lambda a: torch.from_numpy(a).double()
    a, b = rnd(rs, M, K), rnd(rs, K, N)
    gate = (rnd(rs, M, N) > 0).astype(np.float32)
    cs0 = rnd(rs, N)
    C = torch.empty(M, N, device=DEV)
    cs = g(cs0.copy())
    ad, bd, gd = g(a), g(b), g(gate)
    ops.gemm(L.GEMM_NN, [dict(M=M, A=[ad], B=[bd], C=C, gate=gd, colsum=cs)], N, K, K, N, N, ldgate=N, gate_scale=1.5)
    ref = torch.where(td(gate) > 0, (td(a) @ td(b)) * 1.5, torch.zeros(M, N, dtype=torch.float64))
    assert rel_err(C.cpu().numpy(), ref.numpy()) < 1e-5
    want = td(cs0) + ref.sum(0)
    assert rel_err(cs.cpu().numpy(), want.numpy()) < 2e-5


def test_gemm_groups_segments_epilogue():
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    from oracle import dropout_rng
    rs = np.random.RandomState(5)
    M, N, K = 200, 96, 64
    # three K-segments accumulated + bias + relu + dropout + residual
    As = [rnd(rs, M, K) for _ in range(3)]
    Bs = [rnd(rs, N, K) for _ in range(3)]
    bias, res = rnd(rs, N), rnd(rs, M, N)
    seed, site, p = 12345678901234, 1, 0.25
    C = torch.empty(M, N, device=DEV)
    ops.gemm(L.GEMM_NT, [dict(M=M, A=[g(a) for a in As], B=[g(b) for b in Bs], C=C, bias=g(bias), residual=g(res))],
             N, K, K, K, N, nseg=3, relu=True, drop=(p, seed, site), ldres=N)
    acc = sum(torch.from_numpy(a).double() @ torch.from_numpy(b).double().t() for a, b in zip(As, Bs))
    ref = torch.relu(acc + torch.from_numpy(bias).double())
    ref = ref * torch.from_numpy(dropout_rng.scaled_mask(seed, site, (M, N), p)).double() + torch.from_numpy(res).double()
    assert rel_err(C.cpu().numpy(), ref.numpy()) < 1e-5
    # three groups with different M, gate epilogue on one of them
    Ms = [70, 200, 33]
    A = [rnd(rs, m, K) for m in Ms]
    B = [rnd(rs, N, K) for _ in Ms]
    gate = rnd(rs, Ms[1], N)
    Cs = [torch.empty(m, N, device=DEV) for m in Ms]
    groups = [dict(M=m, A=[g(a)], B=[g(b)], C=c) for m, a, b, c in zip(Ms, A, B, Cs)]
    ops.gemm(L.GEMM_NT, groups, N, K, K, K, N)
    for a, b, c in zip(A, B, Cs):
        assert rel_err(c.cpu().numpy(), (torch.from_numpy(a).double() @ torch.from_numpy(b).double().t()).numpy()) < 1e-5
    C1 = torch.empty(Ms[1], N, device=DEV)
    ops.gemm(L.GEMM_NT, [dict(M=Ms[1], A=[g(A[1])], B=[g(B[1])], C=C1, gate=g(gate))], N, K, K, K, N,
             gate_scale=2.0, ldgate=N)
    ref = (torch.from_numpy(A[1]).double() @ torch.from_numpy(B[1]).double().t()) * (torch.from_numpy(gate) > 0).double() * 2.0
    assert rel_err(C1.cpu().numpy(), ref.numpy()) < 1e-5


def test_gemm_split_k_accumulates_onto_c():
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    rs = np.random.RandomState(6)
    K, M, N = 6400, 256, 256
    A, B, C0 = rnd(rs, K, M), rnd(rs, K, N), rnd(rs, M, N)
    ref = torch.from_numpy(C0).double() + torch.from_numpy(A).double().t() @ torch.from_numpy(B).double()
    for kw in (dict(split_k=16), dict(accumulate=True)):     # split_k > 1 is the legacy spelling of accumulate
        C = g(C0)
        ops.gemm(L.GEMM_TN, [dict(M=M, A=[g(A)], B=[g(B)], C=C)], N, K, M, N, N, **kw)
        assert rel_err(C.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize('tile', [64, 128])
@pytest.mark.parametrize('wgs,min_units', [(7, 1), (33, 1), (256, 2), (1024, 1), (0, 4)])
def test_gemm_stream_k_partial_tiles(gemm_tuning, tile, wgs, min_units):
    """Stream-K schedule: output tiles cut along K at arbitrary unit boundaries (forced through the tuning
    knobs), grouped problems with different M, several K-segments, every epilogue term.  The result must
    not depend on the cut, and must be bitwise identical from launch to launch (fixed summation order)."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    from oracle import dropout_rng
    gemm_tuning(tile=tile, sk=2, min_units=min_units, wgs=wgs or None)
    rs = np.random.RandomState(100 + tile + wgs)
    td = lambda a: torch.from_numpy(a).double()
    # NT, three groups, two K segments, bias + residual
    Ms, N, K = [300, 77, 130], 200, 160
    A = [[rnd(rs, m, K) for _ in range(2)] for m in Ms]
    B = [[rnd(rs, N, K) for _ in range(2)] for _ in Ms]
    bias = [rnd(rs, N) for _ in Ms]
    res = [rnd(rs, m, N) for m in Ms]
    outs = []
    for rep in range(3):
        Cs = [torch.full((m, N), float('nan'), device=DEV) for m in Ms]
        groups = [dict(M=m, A=[g(x) for x in a], B=[g(x) for x in b], C=c, bias=g(bi), residual=g(r))
                  for m, a, b, c, bi, r in zip(Ms, A, B, Cs, bias, res)]
        ops.gemm(L.GEMM_NT, groups, N, K, K, K, N, nseg=2, ldres=N)
        outs.append([c.cpu().numpy() for c in Cs])
    for i in range(len(Ms)):
        ref = sum(td(a) @ td(b).t() for a, b in zip(A[i], B[i])) + td(bias[i]) + td(res[i])
        assert rel_err(outs[0][i], ref.numpy()) < 1e-5
        assert np.array_equal(outs[0][i], outs[1][i]) and np.array_equal(outs[0][i], outs[2][i])
    # NN with relu + dropout epilogue on a cut tile
    M, N, K = 333, 96, 512
    a, b = rnd(rs, M, K), rnd(rs, K, N)
    seed, site, p = 987654321, 3, 0.3
    C = torch.empty(M, N, device=DEV)
    ops.gemm(L.GEMM_NN, [dict(M=M, A=[g(a)], B=[g(b)], C=C)], N, K, K, N, N, relu=True, drop=(p, seed, site))
    ref = torch.relu(td(a) @ td(b)) * td(dropout_rng.scaled_mask(seed, site, (M, N), p))
    assert rel_err(C.cpu().numpy(), ref.numpy()) < 1e-5
    # TN weight gradient accumulating onto C, long reduction, two groups
    K, Ms, N = 1888, [96, 160], 72
    A = [rnd(rs, K, m) for m in Ms]
    B = [rnd(rs, K, N) for _ in Ms]
    C0 = [rnd(rs, m, N) for m in Ms]
    for m, x, y, c0 in zip(Ms, A, B, C0):
        c = g(c0)
        ops.gemm(L.GEMM_TN, [dict(M=m, A=[g(x)], B=[g(y)], C=c)], N, K, m, N, N, accumulate=True)
        assert rel_err(c.cpu().numpy(), (td(c0) + td(x).t() @ td(y)).numpy()) < 1e-5
    # odd K (generic guarded path) cut along K
    M, N, K = 150, 130, 203
    a, b = rnd(rs, M, K), rnd(rs, N, K)
    C = torch.empty(M, N, device=DEV)
    ops.gemm(L.GEMM_NT, [dict(M=M, A=[g(a)], B=[g(b)], C=C)], N, K, K, K, N)
    assert rel_err(C.cpu().numpy(), (td(a) @ td(b).t()).numpy()) < 1e-5


def test_gemm_hybrid_schedule_whole_tiles_plus_streamed_tail():
    """256 < tiles < 1024 with a ragged last layer (the 6400 x 512 products: 800 tiles): the library computes
    floor(tiles / 256) * 256 tiles whole and streams the tail through short workgroups.  Every epilogue term,
    grouped problems (tile numbering across groups), bitwise repeatability."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    from oracle import dropout_rng
    rs = np.random.RandomState(77)
    td = lambda a: torch.from_numpy(a).double()
    # NT 1700 x 1024 x 512: 27 x 16 = 432 tiles of 64^2, 16 K-tiles -> 256 whole + 176 streamed
    M, N, K = 1700, 1024, 512
    a, b, bias, res = rnd(rs, M, K), rnd(rs, N, K), rnd(rs, N), rnd(rs, M, N)
    seed, site, p = 5551212, 2, 0.2
    outs = []
    for _ in range(3):
        C = torch.full((M, N), float('nan'), device=DEV)
        ops.gemm(L.GEMM_NT, [dict(M=M, A=[g(a)], B=[g(b)], C=C, bias=g(bias), residual=g(res))], N, K, K, K, N,
                 relu=True, drop=(p, seed, site), ldres=N)
        outs.append(C.cpu().numpy())
    ref = torch.relu(td(a) @ td(b).t() + td(bias)) * td(dropout_rng.scaled_mask(seed, site, (M, N), p)) + td(res)
    assert rel_err(outs[0], ref.numpy()) < 1e-5
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    # NN, three groups of different M (300 + 1000 + 777 rows -> 5 + 16 + 13 row tiles x 12 = 408 tiles), two K segments
    Ms, N, K = [300, 1000, 777], 768, 256
    A = [[rnd(rs, m, K) for _ in range(2)] for m in Ms]
    B = [[rnd(rs, K, N) for _ in range(2)] for _ in Ms]
    gate = [rnd(rs, m, N) for m in Ms]
    Cs = [torch.full((m, N), float('nan'), device=DEV) for m in Ms]
    ops.gemm(L.GEMM_NN, [dict(M=m, A=[g(x) for x in aa], B=[g(x) for x in bb], C=c, gate=g(gt))
                         for m, aa, bb, c, gt in zip(Ms, A, B, Cs, gate)], N, K, K, N, N, nseg=2, gate_scale=1.25, ldgate=N)
    for aa, bb, gt, c in zip(A, B, gate, Cs):
        ref = sum(td(x) @ td(y) for x, y in zip(aa, bb)) * (td(gt) > 0).double() * 1.25
        assert rel_err(c.cpu().numpy(), ref.numpy()) < 1e-5


# ----------------------------------------------------------------------------- LayerNorm & friends
@pytest.mark.parametrize('M,d', [(15, 128), (6400, 512), (64, 1024), (7, 2048), (33, 36)])
def test_layernorm(M, d):
    from mmnas_amd import ops
    from oracle import mmnas_oracle as O
    rs = np.random.RandomState(d + M)
    x, a, b, gy = rnd(rs, M, d) * 2 + 0.3, 1 + 0.2 * rnd(rs, d), 0.1 * rnd(rs, d), rnd(rs, M, d)
    xd, ad, bd = g(x).requires_grad_(True), g(a).requires_grad_(True), g(b).requires_grad_(True)
    y = ops.layer_norm(xd, ad, bd)
    y.backward(g(gy))
    X, A, Bb = (torch.from_numpy(v).double() for v in (x, a, b))
    yr = O.layer_norm(X, A, Bb)
    dx, da, db = O.layer_norm_backward(X, A, torch.from_numpy(gy).double())
    assert rel_err(y.detach().cpu().numpy(), yr.numpy()) < 1e-5
    assert rel_err(xd.grad.cpu().numpy(), dx.numpy()) < 1e-4
    assert rel_err(ad.grad.cpu().numpy(), da.numpy()) < 1e-4
    assert rel_err(bd.grad.cpu().numpy(), db.numpy()) < 1e-4


def test_layernorm_bwd_dropout_and_colsum_outputs():
    import mmnas_amd._lib as L
    from oracle import dropout_rng, mmnas_oracle as O
    rs = np.random.RandomState(3)
    M, d, p, seed = 50, 256, 0.2, 99
    x, a, gy = rnd(rs, M, d), 1 + 0.1 * rnd(rs, d), rnd(rs, M, d)
    dx, dd = torch.empty(M, d, device=DEV), torch.empty(M, d, device=DEV)
    dab = torch.zeros(3, d, device=DEV)
    xd, ad, gd = g(x), g(a), g(gy)  # keep the device tensors alive across the asynchronous launch
    for use_ws in (False, True):
        dab.zero_()
        ws = torch.empty(L.lib().mmnas_layernorm_bwd_ws_floats(M, d), device=DEV) if use_ws else None
        L.check(L.lib().mmnas_layernorm_bwd(L.fptr(xd), L.fptr(ad), L.fptr(gd), L.fptr(dx), L.fptr(dab[0]),
                                            L.fptr(dab[1]), L.fptr(dd), L.fptr(dab[2]), L.fptr(ws), p, seed, 1, M, d,
                                            1e-6, L.stream()))
        _check_ln_bwd(O, dropout_rng, x, a, gy, dx, dd, dab, seed, p, M, d)


def _check_ln_bwd(O, dropout_rng, x, a, gy, dx, dd, dab, seed, p, M, d):
    rdx, _, _ = O.layer_norm_backward(torch.from_numpy(x).double(), torch.from_numpy(a).double(), torch.from_numpy(gy).double())
    rdd = rdx * torch.from_numpy(dropout_rng.scaled_mask(seed, 1, (M, d), p)).double()
    assert rel_err(dx.cpu().numpy(), rdx.numpy()) < 1e-4
    assert rel_err(dd.cpu().numpy(), rdd.numpy()) < 1e-4
    assert rel_err(dab[2].cpu().numpy(), rdd.sum(0).numpy()) < 1e-4


def test_colsum_eltwise_glu_dropadd():
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    from oracle import dropout_rng, mmnas_oracle as O
    rs = np.random.RandomState(8)
    x = rnd(rs, 777, 300)
    out = torch.zeros(300, device=DEV)
    xdev = g(x)
    L.check(L.lib().mmnas_colsum(L.fptr(xdev), L.fptr(out), 777, 300, 300, L.stream()))
    assert rel_err(out.cpu().numpy(), x.astype(np.float64).sum(0)) < 1e-5
    xt = torch.from_numpy(x).double().requires_grad_(True)
    gy = rnd(rs, 777, 300)
    refs = {0: xt * 0., 1: torch.relu(xt), 2: torch.nn.functional.leaky_relu(xt, 0.01), 3: O.gelu_tanh(xt)}
    for kind, r in refs.items():
        xd = g(x).requires_grad_(True)
        y = ops.eltwise(xd, kind)
        y.backward(g(gy))
        xt.grad = None
        r.backward(torch.from_numpy(gy).double())
        assert rel_err(y.detach().cpu().numpy(), r.detach().numpy()) < 1e-5, kind
        assert rel_err(xd.grad.cpu().numpy(), xt.grad.numpy()) < 1e-5, kind
    # GLU with relu + dropout
    h = rnd(rs, 60, 256)
    hd = g(h).requires_grad_(True)
    seed, p = 4242, 0.3
    y = ops.glu(hd, relu=True, drop_p=p, seed=seed, site=0)
    gy = rnd(rs, 60, 128)
    y.backward(g(gy))
    ht = torch.from_numpy(h).double().requires_grad_(True)
    a, b = ht.chunk(2, -1)
    r = torch.relu(a * torch.sigmoid(b)) * torch.from_numpy(dropout_rng.scaled_mask(seed, 0, (60, 128), p)).double()
    r.backward(torch.from_numpy(gy).double())
    assert rel_err(y.detach().cpu().numpy(), r.detach().numpy()) < 1e-5
    assert rel_err(hd.grad.cpu().numpy(), ht.grad.numpy()) < 1e-5
    # drop_add
    xx, rr = rnd(rs, 40, 64), rnd(rs, 40, 64)
    z = ops.drop_add(g(xx), g(rr), 0.5, 77, 1)
    ref = rr + xx * dropout_rng.scaled_mask(77, 1, (40, 64), 0.5)
    assert rel_err(z.cpu().numpy(), ref) < 1e-6


# ----------------------------------------------------------------------------- relation bias
@pytest.mark.parametrize('B,Sq,Sk,R,H', [(2, 7, 7, 64, 2), (2, 100, 100, 64, 8), (1, 9, 9, 64, 16), (3, 5, 5, 64, 1),
                                         (2, 6, 6, 32, 4), (1, 5, 5, 64, 32)])
def test_rel_bias(B, Sq, Sk, R, H):
    import mmnas_amd._lib as L
    rs = np.random.RandomState(B * 100 + Sq + H)
    rel = np.maximum(rnd(rs, B, Sq, Sk, R), 0)
    Wr, br = rnd(rs, H, R) / 8, 0.1 * rnd(rs, H)
    gb = rnd(rs, B, H, Sk, Sq)
    reld, Wd, bd = g(rel), g(Wr), g(br)
    biasT = torch.empty(B, H, Sk, Sq, device=DEV)
    L.check(L.lib().mmnas_rel_bias_fwd(L.fptr(reld), L.fptr(Wd), L.fptr(bd), L.fptr(biasT), B, Sq, Sk, R, H, L.stream()))
    relt = torch.from_numpy(rel).double().requires_grad_(True)
    Wt = torch.from_numpy(Wr).double().requires_grad_(True)
    bt = torch.from_numpy(br).double().requires_grad_(True)
    r = torch.relu(relt @ Wt.t() + bt)                      # [B,Sq,Sk,H]
    bias = torch.log(torch.clamp(r, min=1e-6)).permute(0, 3, 2, 1)  # -> [B,H,Sk,Sq]
    assert rel_err(biasT.cpu().numpy(), bias.detach().numpy()) < TOL
    bias.backward(torch.from_numpy(gb).double())
    drel = torch.empty_like(reld)
    dW, db = torch.zeros(H, R, device=DEV), torch.zeros(H, device=DEV)
    gbd = g(gb)
    L.check(L.lib().mmnas_rel_bias_bwd(L.fptr(reld), L.fptr(Wd), L.fptr(bd), L.fptr(gbd), L.fptr(drel), L.fptr(dW),
                                       L.fptr(db), 0, B, Sq, Sk, R, H, L.stream()))
    assert rel_err(drel.cpu().numpy(), relt.grad.numpy()) < TOL
    # dWr/dbr sum B*Sq*Sk random-sign terms dbias/r (1/r amplified near the clamp): fp32 summation of
    # 20000 such terms against an fp64 reference carries ~1e-3 of cancellation noise
    wtol = 3e-3 if B * Sq * Sk > 5000 else TOL
    assert rel_err(dW.cpu().numpy(), Wt.grad.numpy()) < wtol
    assert rel_err(db.cpu().numpy(), bt.grad.numpy()) < wtol


@pytest.mark.parametrize('B,Sq,Sk,C,H', [(2, 7, 7, 4, 2), (2, 100, 100, 4, 8), (3, 5, 9, 3, 4), (1, 70, 3, 4, 16),
                                         (2, 14, 14, 3, 32)])
def test_rel_fused_lazy_handle(B, Sq, Sk, C, H):
    """bias and parameter gradients straight from the raw relation tensor == linear_y_rel -> relu -> rel bias."""
    import mmnas_amd._lib as L
    rs = np.random.RandomState(B * 131 + Sq + H + C)
    R = 64
    raw = rnd(rs, B, Sq, Sk, C)
    raw[:, Sq // 2:, :, :] *= (rs.uniform(size=(B, Sq - Sq // 2, Sk, 1)) < 0.7)   # zero-padded rows, as the loader makes
    Wy, by = rnd(rs, R, C) / 2, 0.1 * rnd(rs, R)
    Wr, br = rnd(rs, H, R) / 8, 0.1 * rnd(rs, H)
    gb = rnd(rs, B, H, Sk, Sq)
    rawd, Wyd, byd, Wrd, brd, gbd = g(raw), g(Wy), g(by), g(Wr), g(br), g(gb)
    assert L.lib().mmnas_rel_fused_supported(C, R, H) == 1
    biasT = torch.empty(B, H, Sk, Sq, device=DEV)
    L.check(L.lib().mmnas_rel_fused_fwd(L.fptr(rawd), L.fptr(Wyd), L.fptr(byd), L.fptr(Wrd), L.fptr(brd), L.fptr(biasT),
                                        B, Sq, Sk, C, R, H, L.stream()))
    T = lambda a: torch.from_numpy(a).double().requires_grad_(True)
    Wyt, byt, Wrt, brt = T(Wy), T(by), T(Wr), T(br)
    rel = torch.relu(torch.from_numpy(raw).double() @ Wyt.t() + byt)
    r = torch.relu(rel @ Wrt.t() + brt)
    bias = torch.log(torch.clamp(r, min=1e-6)).permute(0, 3, 2, 1)
    assert rel_err(biasT.cpu().numpy(), bias.detach().numpy()) < TOL
    bias.backward(torch.from_numpy(gb).double())
    dWy, dby = torch.zeros(R, C, device=DEV), torch.zeros(R, device=DEV)
    dWr, dbr = torch.zeros(H, R, device=DEV), torch.zeros(H, device=DEV)
    ws = torch.empty(L.lib().mmnas_rel_fused_bwd_ws_floats(B, Sq, Sk), device=DEV)
    L.check(L.lib().mmnas_rel_fused_bwd(L.fptr(rawd), L.fptr(Wyd), L.fptr(byd), L.fptr(Wrd), L.fptr(brd), L.fptr(gbd),
                                        L.fptr(dWy), L.fptr(dby), L.fptr(dWr), L.fptr(dbr), L.fptr(ws), B, Sq, Sk, C, R, H,
                                        L.stream()))
    # 1/r-amplified random-sign sums over 20000 elements (see test_rel_bias); here they are additionally
    # accumulated as one sequential fp32 fma chain per workgroup on the MFMA
    wtol = 6e-3 if B * Sq * Sk > 5000 else TOL
    assert rel_err(dWr.cpu().numpy(), Wrt.grad.numpy()) < wtol
    assert rel_err(dbr.cpu().numpy(), brt.grad.numpy()) < wtol
    assert rel_err(dWy.cpu().numpy(), Wyt.grad.numpy()) < wtol
    assert rel_err(dby.cpu().numpy(), byt.grad.numpy()) < wtol


@pytest.mark.parametrize('B,S,C,H,lens', [(4, 100, 4, 4, [100, 37, 10, 1]), (3, 100, 4, 8, [64, 100, 33]), (5, 14, 3, 4, [14, 3, 7, 1, 9]),
                                         (2, 36, 4, 16, [36, 20])])
def test_rel_fused_ragged(B, S, C, H, lens):
    """mmnas_rel_fused_fwd_ragged / _bwd_ragged: the lazy relation handle over the valid n_b x n_b corner of every sample.
    Forward equals the dense call there (and touches nothing else); backward equals the dense call on a bias gradient that
    is zero outside the corners -- what the attention backward of a padded batch produces."""
    import mmnas_amd._lib as L
    rs = np.random.RandomState(B * 17 + S + H + C)
    R = 64
    raw = rnd(rs, B, S, S, C)
    Wy, by = rnd(rs, R, C) / 2, 0.1 * rnd(rs, R)
    Wr, br = rnd(rs, H, R) / 8, 0.1 * rnd(rs, H)
    gb = rnd(rs, B, H, S, S)
    valid = np.zeros((B, 1, S, S), np.float32)
    for b, n in enumerate(lens):
        valid[b, :, :n, :n] = 1
    gb_dense = gb * valid
    rawd, Wyd, byd, Wrd, brd = g(raw), g(Wy), g(by), g(Wr), g(br)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    toff = np.concatenate([[0], np.cumsum([(n * n + 31) // 32 for n in lens])]).astype(np.int32)
    offd, toffd = g(off), g(toff)
    dense = torch.empty(B, H, S, S, device=DEV)
    L.check(L.lib().mmnas_rel_fused_fwd(L.fptr(rawd), L.fptr(Wyd), L.fptr(byd), L.fptr(Wrd), L.fptr(brd), L.fptr(dense), B, S, S, C, R, H, L.stream()))
    rag = torch.full((B, H, S, S), 12345.0, device=DEV)
    L.check(L.lib().mmnas_rel_fused_fwd_ragged(L.fptr(rawd), L.fptr(Wyd), L.fptr(byd), L.fptr(Wrd), L.fptr(brd), L.fptr(rag), B, S, C, R, H,
                                               L.ptr(offd), L.stream()))
    v = torch.from_numpy(valid).to(DEV).bool().expand(B, H, S, S)
    assert torch.equal(rag[v], dense[v]) and bool((rag[~v] == 12345.0).all())
    outs = []
    for ragged in (False, True):
        dWy, dby = torch.zeros(R, C, device=DEV), torch.zeros(R, device=DEV)
        dWr, dbr = torch.zeros(H, R, device=DEV), torch.zeros(H, device=DEV)
        ws = torch.empty(L.lib().mmnas_rel_fused_bwd_ws_floats(B, S, S), device=DEV)
        if ragged:
            gbd = g(np.where(valid > 0, gb, np.nan).astype(np.float32))      # outside the corners: never read
            L.check(L.lib().mmnas_rel_fused_bwd_ragged(L.fptr(rawd), L.fptr(Wyd), L.fptr(byd), L.fptr(Wrd), L.fptr(brd), L.fptr(gbd),
                                                       L.fptr(dWy), L.fptr(dby), L.fptr(dWr), L.fptr(dbr), L.fptr(ws), B, S, C, R, H,
                                                       L.ptr(offd), L.ptr(toffd), int(toff[-1]), L.stream()))
        else:
            gbd = g(gb_dense)
            L.check(L.lib().mmnas_rel_fused_bwd(L.fptr(rawd), L.fptr(Wyd), L.fptr(byd), L.fptr(Wrd), L.fptr(brd), L.fptr(gbd),
                                                L.fptr(dWy), L.fptr(dby), L.fptr(dWr), L.fptr(dbr), L.fptr(ws), B, S, S, C, R, H, L.stream()))
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in (dWy, dby, dWr, dbr)])
    for a, c in zip(outs[1], outs[0]):
        assert np.isfinite(a).all()
        assert rel_err(a, c) < 2e-4         # (another summation order over the same terms)


@pytest.mark.parametrize('B,S,C,H,n_ops,lens', [
    (2, 7, 4, 2, 3, None), (2, 100, 4, 4, 6, None), (2, 100, 4, 8, 5, None), (3, 14, 3, 4, 1, None), (1, 33, 4, 16, 3, None),
    (2, 40, 4, 4, 18, None), (2, 23, 4, 1, 2, None), (4, 23, 4, 4, 9, (23, 1, 7, 16)), (3, 100, 4, 8, 4, (100, 37, 64)), (2, 9, 4, 32, 2, None)])
def test_rel_multi_all_relation_operators_in_one_launch(B, S, C, H, n_ops, lens):
    """mmnas_rel_multi_fwd / _bwd (relmulti.hip): the relation bias of n_ops RelSelfAtt operators that share the stem layer
    linear_y_rel, each with its own linear_r (modules.py:231-235, hygr_vqa.py:111) -- the hidden layer computed once per
    element, every operator's bias written / bias gradient read in the same launch, dWy / dby contracted once.  Against the
    float64 restatement of the reference arithmetic (forward: on max(r, 1e-6) itself, the log's argument -- an r next to
    zero carries its cancellation error into the log at any precision) and against the per-operator kernels of
    relfused.hip (dense and ragged)."""
    import ctypes as C_
    import mmnas_amd._lib as L
    lib = L.lib()
    rs = np.random.RandomState(B * 31 + S + 7 * H + C + n_ops)
    R = 64
    assert lib.mmnas_rel_multi_supported(C, R, H) == 1
    raw = rnd(rs, B, S, S, C)
    Wy, by = rnd(rs, R, C) / 2, 0.1 * rnd(rs, R)
    Wrs = [rnd(rs, H, R) / 8 for _ in range(n_ops)]
    brs = [0.1 * rnd(rs, H) for _ in range(n_ops)]
    gbs = [rnd(rs, B, H, S, S) for _ in range(n_ops)]
    valid = np.ones((B, 1, S, S), np.float32)
    offd = toffd = None
    ntiles = 0
    if lens is not None:
        valid[:] = 0
        for b, n in enumerate(lens):
            valid[b, :, :n, :n] = 1
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        toff = np.concatenate([[0], np.cumsum([(n * n + 31) // 32 for n in lens])]).astype(np.int32)
        offd, toffd, ntiles = g(off), g(toff), int(toff[-1])
    vmask = torch.from_numpy(valid).bool().expand(B, H, S, S)
    rawd, Wyd, byd = g(raw), g(Wy), g(by)
    Wrd, brd = [g(w) for w in Wrs], [g(b) for b in brs]
    # d log(max(r, 1e-6)) / dr = 1 / r jumps to 0 at the clamp: an element whose pre-activation sits next to zero is on either
    # side of it depending on the last bit, and weighs up to 1e6 in the sums.  The test keeps such elements out of the
    # gradient (zero bias gradient inside the band |pre| < 0.05) so that what is left is well-conditioned and can be held to
    # a tight bound; elements safely BELOW the clamp keep a non-zero bias gradient (their contribution must be exactly zero).
    rel64 = torch.relu(torch.from_numpy(raw).double() @ torch.from_numpy(Wy).double().t() + torch.from_numpy(by).double())
    for i in range(n_ops):
        pre = (rel64 @ torch.from_numpy(Wrs[i]).double().t() + torch.from_numpy(brs[i]).double()).permute(0, 3, 2, 1).numpy()
        gbs[i] = np.where(np.abs(pre) < 0.05, 0.0, gbs[i]).astype(np.float32)
    # outside the valid corners the bias gradient is never read: poison it
    gbd = [g(np.where(valid > 0, gb, np.nan).astype(np.float32)) for gb in gbs]
    bias = [torch.full((B, H, S, S), 12345.0, device=DEV) for _ in range(n_ops)]
    dWr, dbr = [torch.zeros(H, R, device=DEV) for _ in range(n_ops)], [torch.zeros(H, device=DEV) for _ in range(n_ops)]
    dWy, dby = torch.zeros(R, C, device=DEV), torch.zeros(R, device=DEV)
    ws = torch.empty(lib.mmnas_rel_multi_bwd_ws_floats(B, S), device=DEV)
    m = L.RelMulti()
    m.B, m.S, m.C, m.R, m.H, m.n_ops = B, S, C, R, H, n_ops
    m.raw, m.Wy, m.by, m.dWy, m.dby, m.ws = L.fptr(rawd), L.fptr(Wyd), L.fptr(byd), L.fptr(dWy), L.fptr(dby), L.fptr(ws)
    for i in range(n_ops):
        m.Wr[i], m.br[i], m.biasT[i], m.dbiasT[i] = L.fptr(Wrd[i]), L.fptr(brd[i]), L.fptr(bias[i]), L.fptr(gbd[i])
        m.dWr[i], m.dbr[i] = L.fptr(dWr[i]), L.fptr(dbr[i])
    if lens is not None:
        m.off, m.tile_off, m.ntiles = L.ptr(offd), L.ptr(toffd), ntiles
    L.check(lib.mmnas_rel_multi_fwd(C_.byref(m), L.stream()))
    L.check(lib.mmnas_rel_multi_bwd(C_.byref(m), L.stream()))
    torch.cuda.synchronize()
    # float64 restatement
    T = lambda a: torch.from_numpy(a).double().requires_grad_(True)
    Wyt, byt = T(Wy), T(by)
    rel = torch.relu(torch.from_numpy(raw).double() @ Wyt.t() + byt)
    vm = torch.from_numpy(valid).double()
    refs = []
    for i in range(n_ops):
        Wrt, brt = T(Wrs[i]), T(brs[i])
        r = torch.clamp(torch.relu(rel @ Wrt.t() + brt), min=1e-6).permute(0, 3, 2, 1)      # [B, H, S_k, S_q]
        (torch.log(r) * torch.from_numpy(gbs[i]).double() * vm).sum().backward(retain_graph=True)
        refs.append((r.detach(), Wrt.grad, brt.grad))
    wtol = 1e-4
    for i in range(n_ops):
        got = bias[i].cpu()
        assert bool((got[~vmask] == 12345.0).all()), 'the forward wrote outside the valid corners'
        rr = torch.exp(torch.where(vmask, got, torch.zeros(())).double())
        scale = float(refs[i][0].abs().max())
        assert float(((rr - refs[i][0]).abs() * vm).max()) <= 2e-5 * scale, i
        assert rel_err(dWr[i].cpu().numpy(), refs[i][1].numpy()) < wtol, i
        assert rel_err(dbr[i].cpu().numpy(), refs[i][2].numpy()) < wtol, i
    assert rel_err(dWy.cpu().numpy(), Wyt.grad.numpy()) < wtol
    assert rel_err(dby.cpu().numpy(), byt.grad.numpy()) < wtol
    # ... and the per-operator kernels: same numbers to round-off (another summation order over the same terms)
    p_dWy, p_dby = torch.zeros(R, C, device=DEV), torch.zeros(R, device=DEV)
    ws1 = torch.empty(lib.mmnas_rel_fused_bwd_ws_floats(B, S, S), device=DEV)
    for i in range(n_ops):
        pb = torch.full((B, H, S, S), 12345.0, device=DEV)
        p_dWr, p_dbr = torch.zeros(H, R, device=DEV), torch.zeros(H, device=DEV)
        if lens is None:
            L.check(lib.mmnas_rel_fused_fwd(L.fptr(rawd), L.fptr(Wyd), L.fptr(byd), L.fptr(Wrd[i]), L.fptr(brd[i]), L.fptr(pb), B, S, S, C, R, H, L.stream()))
            L.check(lib.mmnas_rel_fused_bwd(L.fptr(rawd), L.fptr(Wyd), L.fptr(byd), L.fptr(Wrd[i]), L.fptr(brd[i]), L.fptr(gbd[i]),
                                            L.fptr(p_dWy), L.fptr(p_dby), L.fptr(p_dWr), L.fptr(p_dbr), L.fptr(ws1), B, S, S, C, R, H, L.stream()))
        else:
            L.check(lib.mmnas_rel_fused_fwd_ragged(L.fptr(rawd), L.fptr(Wyd), L.fptr(byd), L.fptr(Wrd[i]), L.fptr(brd[i]), L.fptr(pb), B, S, C, R, H,
                                                   L.ptr(offd), L.stream()))
            L.check(lib.mmnas_rel_fused_bwd_ragged(L.fptr(rawd), L.fptr(Wyd), L.fptr(byd), L.fptr(Wrd[i]), L.fptr(brd[i]), L.fptr(gbd[i]),
                                                   L.fptr(p_dWy), L.fptr(p_dby), L.fptr(p_dWr), L.fptr(p_dbr), L.fptr(ws1), B, S, C, R, H,
                                                   L.ptr(offd), L.ptr(toffd), ntiles, L.stream()))
        torch.cuda.synchronize()
        vd = vmask.to(DEV)
        a, c = torch.exp(torch.where(vd, bias[i], 0.0).double()), torch.exp(torch.where(vd, pb, 0.0).double())
        assert float((a - c).abs().max()) <= 2e-5 * float(c.abs().max())
        assert rel_err(dWr[i].cpu().numpy(), p_dWr.cpu().numpy()) < 3e-4
        assert rel_err(dbr[i].cpu().numpy(), p_dbr.cpu().numpy()) < 3e-4
    assert rel_err(dWy.cpu().numpy(), p_dWy.cpu().numpy()) < 3e-4
    assert rel_err(dby.cpu().numpy(), p_dby.cpu().numpy()) < 3e-4


def test_rel_multi_refuses_what_it_does_not_cover():
    import ctypes as C_
    import mmnas_amd._lib as L
    lib = L.lib()
    assert lib.mmnas_rel_multi_supported(4, 64, 3) == 0 and lib.mmnas_rel_multi_supported(4, 32, 4) == 0 and lib.mmnas_rel_multi_supported(5, 64, 4) == 0
    m = L.RelMulti()
    m.B, m.S, m.C, m.R, m.H, m.n_ops = 2, 5, 4, 64, 3, 1
    t = torch.zeros(4096, device=DEV)
    m.raw = m.Wy = m.by = L.fptr(t)
    with pytest.raises(L.MMNasHipError):
        L.check(lib.mmnas_rel_multi_fwd(C_.byref(m), L.stream()))
    m.H, m.n_ops = 4, 0
    with pytest.raises(L.MMNasHipError):
        L.check(lib.mmnas_rel_multi_fwd(C_.byref(m), L.stream()))


# ----------------------------------------------------------------------------- stem / head helpers
@pytest.mark.parametrize('shape', [(64, 100, 2048), (3, 7, 5), (2, 9, 36), (1, 1, 4)])
def test_row_is_zero_is_make_mask(shape):
    """mmnas_row_is_zero == (sum |f| == 0) of hygr_vqa.py:121-122, including -0.0 rows and NaN rows."""
    from mmnas_amd import ops
    rs = np.random.RandomState(sum(shape))
    f = rnd(rs, *shape)
    f[rs.uniform(size=shape[:-1]) < 0.4] = 0.0
    if shape[0] > 1:
        f[0, 0] = -0.0
        f[1, 0] = 0.0
        f[1, 0, -1] = np.nan
    ft = torch.from_numpy(f)
    ref = (ft.abs().sum(-1) == 0)
    got = ops.row_is_zero(ft.to(DEV)).cpu()
    assert got.dtype == torch.bool and torch.equal(got, ref)


def test_relation_embedding_batched():
    """loader-side box-geometry features (load_data_vqa.py:224-239) computed batched on the GPU, zero padded
    to S, against the per-sample restatement in the oracle (parity unpinned for this function, see oracle)."""
    from mmnas_amd import ops
    from oracle import mmnas_oracle as O
    rs = np.random.RandomState(31)
    B, S = 5, 100
    x1, y1 = rs.uniform(0, 500, (B, S)), rs.uniform(0, 400, (B, S))
    bw, bh = rs.uniform(1, 300, (B, S)), rs.uniform(1, 200, (B, S))
    bbox = np.stack([x1, y1, x1 + bw, y1 + bh], -1).astype(np.float32)
    bbox[0, 3] = bbox[0, 2]                     # identical boxes: |dcx|/w clamps at 1e-3
    nobj = np.array([100, 37, 1, 64, 10], np.int32)
    got = ops.relation_embedding(g(bbox), torch.from_numpy(nobj)).cpu().numpy()
    for b in range(B):
        n = int(nobj[b])
        ref = np.zeros((S, S, 4), np.float32)
        ref[:n, :n] = O.relation_embedding(torch.from_numpy(bbox[b, :n]).double()).numpy()
        assert rel_err(got[b], ref) < 1e-4   # fp32 centre differences of ~500-pixel coordinates vs the fp64 restatement
    assert np.array_equal(ops.relation_embedding(g(bbox)).cpu().numpy()[0], got[0])   # nobj=None: all S boxes


@pytest.mark.parametrize('B,S,d,G,use_mask', [(64, 100, 512, 1, True), (3, 14, 512, 1, True), (2, 7, 36, 2, True),
                                              (2, 300, 64, 3, False), (1, 1, 8, 1, True)])
def test_attflat_pool(B, S, d, G, use_mask):
    """pooling stage of AttFlat (modules.py:78-84): masked softmax over the sequence + weighted sum, fwd and bwd,
    against the torch expression of the reference in fp64; a fully masked sequence pools uniformly."""
    from mmnas_amd import ops
    rs = np.random.RandomState(B * 7 + S)
    logits, x, gp = rnd(rs, B, S, G), rnd(rs, B, S, d), rnd(rs, B, G * d)
    mask = None
    if use_mask:
        mask = rs.uniform(size=(B, 1, 1, S)) < 0.3
        mask[0] = True                      # everything padded: softmax of equal -1e9 logits is uniform
        mask_t = torch.from_numpy(mask)
    lt, xt = torch.from_numpy(logits).double().requires_grad_(True), torch.from_numpy(x).double().requires_grad_(True)
    att = lt
    if use_mask:
        att = att.masked_fill(mask_t.squeeze(1).squeeze(1).unsqueeze(2), -1e9)
    att = torch.softmax(att, dim=1)
    ref = torch.cat([(att[:, :, g_:g_ + 1] * xt).sum(1) for g_ in range(G)], dim=1)
    ref.backward(torch.from_numpy(gp).double())
    ld, xd = g(logits).requires_grad_(True), g(x).requires_grad_(True)
    out = ops.attflat_pool(ld, xd, mask_t.to(DEV) if use_mask else None)
    out.backward(g(gp))
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    assert rel_err(xd.grad.cpu().numpy(), xt.grad.numpy()) < 1e-5
    assert rel_err(ld.grad.cpu().numpy(), lt.grad.numpy()) < 2e-5


# ----------------------------------------------------------------------------- attention core
def _mha_ref(Q, K, V, mask, biasT, H, dh, dmask=None):
    B, Sq, _ = Q.shape
    Sk = K.shape[1]
    q = Q.reshape(B, Sq, H, dh).permute(0, 2, 1, 3)
    k = K.reshape(B, Sk, H, dh).permute(0, 2, 1, 3)
    v = V.reshape(B, Sk, H, dh).permute(0, 2, 1, 3)
    z = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if biasT is not None:
        z = z + biasT.permute(0, 1, 3, 2)
    if mask is not None:
        z = z.masked_fill(mask.reshape(B, 1, 1, Sk), -1e9)
    a = torch.softmax(z, -1)
    if dmask is not None:
        a = a * dmask
    return (a @ v).permute(0, 2, 1, 3).reshape(B, Sq, H * dh)


@pytest.mark.parametrize('B,H,Sq,Sk,dh,use_mask,use_bias,p', [
    (2, 2, 7, 7, 64, True, False, 0.0), (2, 8, 100, 100, 64, True, True, 0.0), (2, 8, 100, 14, 64, True, False, 0.1),
    (3, 4, 14, 14, 64, True, False, 0.0), (2, 2, 36, 50, 64, True, False, 0.0), (2, 4, 9, 9, 32, True, True, 0.0),
    (2, 8, 9, 15, 16, False, False, 0.0), (2, 2, 9, 9, 128, True, False, 0.0), (2, 1, 7, 12, 256, True, True, 0.0),
    (1, 2, 130, 130, 64, True, False, 0.0), (2, 2, 50, 200, 64, True, True, 0.2), (2, 4, 70, 33, 32, True, False, 0.0),
    # few keys, many queries: query-split dK/dV workgroups, all head-dim chunk widths
    (2, 4, 100, 14, 32, True, False, 0.1), (2, 8, 100, 20, 16, True, True, 0.0), (2, 2, 40, 10, 64, False, False, 0.0),
    (1, 2, 150, 31, 128, True, False, 0.0),
    # round 6: the bf16-pipe cores (d_h = 64, 65..128 keys, <= 128 queries): dropout + bias + mask together, fewer queries than
    # keys, the range's ends (65 / 128 keys; a wave without keys at 96; one query; 128 queries), a fully padded sample
    (2, 4, 100, 100, 64, True, True, 0.1), (2, 2, 36, 100, 64, True, False, 0.1), (3, 1, 128, 65, 64, True, True, 0.0),
    (2, 2, 1, 96, 64, False, False, 0.0), (2, 4, 97, 128, 64, True, False, 0.3), (2, 3, 128, 128, 64, False, True, 0.0)])
def test_mha_core(B, H, Sq, Sk, dh, use_mask, use_bias, p):
    import ctypes as C
    import mmnas_amd._lib as L
    from oracle import dropout_rng
    rs = np.random.RandomState(B + H * 3 + Sq * 5 + Sk * 7 + dh)
    di = H * dh
    Q, K, V, dO = rnd(rs, B, Sq, di), rnd(rs, B, Sk, di), rnd(rs, B, Sk, di), rnd(rs, B, Sq, di)
    mask = np.zeros((B, Sk), np.bool_)
    if use_mask:
        for b in range(1, B):
            mask[b, int(rs.randint(1, Sk)):] = True
        if B > 1:
            mask[B - 1] = True      # fully padded sample: uniform softmax
    biasT = (rnd(rs, B, H, Sk, Sq) * 2) if use_bias else None
    seed = 31337
    Qd, Kd, Vd, dOd = g(Q), g(K), g(V), g(dO)
    m8 = g(mask.astype(np.uint8)) if use_mask else None
    bd = g(biasT) if use_bias else None
    O_ = torch.empty(B, Sq, di, device=DEV)
    stats = torch.empty(B, H, Sq, 2, device=DEV)
    d = L.MhaDesc()
    d.B, d.H, d.Sq, d.Sk, d.dh = B, H, Sq, Sk, dh
    d.ldq = d.ldk = d.ldv = d.ldo = di
    d.Q, d.K, d.V, d.mask, d.biasT, d.O, d.lse = (L.fptr(Qd), L.fptr(Kd), L.fptr(Vd), L.ptr(m8), L.fptr(bd),
                                                 L.fptr(O_), L.fptr(stats))
    d.drop_p, d.drop_site, d.drop_seed = p, 0, seed
    L.check(L.lib().mmnas_mha_core_fwd(C.byref(d), L.stream()))
    Qt, Kt, Vt = (torch.from_numpy(v).double().requires_grad_(True) for v in (Q, K, V))
    bt = torch.from_numpy(biasT).double().requires_grad_(True) if use_bias else None
    dm = torch.from_numpy(dropout_rng.scaled_mask(seed, 0, (B, H, Sq, Sk), p)).double() if p > 0 else None
    ref = _mha_ref(Qt, Kt, Vt, torch.from_numpy(mask) if use_mask else None, bt, H, dh, dm)
    assert rel_err(O_.cpu().numpy(), ref.detach().numpy()) < 1e-5
    ref.backward(torch.from_numpy(dO).double())
    dQ, dK, dV = torch.empty_like(Qd), torch.empty_like(Kd), torch.empty_like(Vd)
    dbT = torch.empty(B, H, Sk, Sq, device=DEV) if use_bias else None
    delta = torch.empty(B, H, Sq, device=DEV)
    d.dO, d.dQ, d.dK, d.dV, d.dbiasT, d.delta = L.fptr(dOd), L.fptr(dQ), L.fptr(dK), L.fptr(dV), L.fptr(dbT), L.fptr(delta)
    L.check(L.lib().mmnas_mha_core_bwd(C.byref(d), L.stream()))
    assert rel_err(dQ.cpu().numpy(), Qt.grad.numpy()) < 1e-4
    assert rel_err(dK.cpu().numpy(), Kt.grad.numpy()) < 1e-4
    assert rel_err(dV.cpu().numpy(), Vt.grad.numpy()) < 1e-4
    if use_bias:
        assert rel_err(dbT.cpu().numpy(), bt.grad.numpy()) < 1e-4


@pytest.mark.parametrize('Sq,Sk,use_bias,p', [(100, 100, True, 0.1), (128, 128, False, 0.0), (36, 65, True, 0.0)])
def test_mha_b16_cores_run_to_run_identical(Sq, Sk, use_bias, p):
    """The bf16-pipe cores have no atomics and no order-dependent sums: six runs of the forward and of the backward on the
    same inputs are bitwise equal.  (Round 6: an image layout tried in the backward gave run-to-run different dQ / dK / dV
    under the kernel's software-pipelined schedule while every single-shape accuracy test of another schedule passed --
    docs/LAB_NOTES.md; this is the guard.)"""
    import ctypes as C
    import mmnas_amd._lib as L
    B, H, dh = 8, 4, 64
    rs = np.random.RandomState(Sq + 3 * Sk)
    di = H * dh
    Qd, Kd, Vd, dOd = g(rnd(rs, B, Sq, di)), g(rnd(rs, B, Sk, di)), g(rnd(rs, B, Sk, di)), g(rnd(rs, B, Sq, di))
    bd = g(rnd(rs, B, H, Sk, Sq) * 2) if use_bias else None
    first = None
    for _ in range(6):
        O_ = torch.empty(B, Sq, di, device=DEV)
        stats = torch.empty(B, H, Sq, 2, device=DEV)
        d = L.MhaDesc()
        d.B, d.H, d.Sq, d.Sk, d.dh = B, H, Sq, Sk, dh
        d.ldq = d.ldk = d.ldv = d.ldo = di
        d.Q, d.K, d.V, d.mask, d.biasT, d.O, d.lse = (L.fptr(Qd), L.fptr(Kd), L.fptr(Vd), L.ptr(None), L.fptr(bd),
                                                     L.fptr(O_), L.fptr(stats))
        d.drop_p, d.drop_site, d.drop_seed = p, 0, 4242
        L.check(L.lib().mmnas_mha_core_fwd(C.byref(d), L.stream()))
        dQ, dK, dV = torch.empty_like(Qd), torch.empty_like(Kd), torch.empty_like(Vd)
        dbT = torch.empty(B, H, Sk, Sq, device=DEV) if use_bias else None
        delta = torch.empty(B, H, Sq, device=DEV)
        d.dO, d.dQ, d.dK, d.dV, d.dbiasT, d.delta = L.fptr(dOd), L.fptr(dQ), L.fptr(dK), L.fptr(dV), L.fptr(dbT), L.fptr(delta)
        L.check(L.lib().mmnas_mha_core_bwd(C.byref(d), L.stream()))
        out = [O_, dQ, dK, dV] + ([dbT] if use_bias else [])
        assert all(bool(torch.isfinite(t).all()) for t in out)
        if first is None:
            first = [t.clone() for t in out]
        else:
            for a, b in zip(first, out):
                assert torch.equal(a, b)


@pytest.mark.parametrize('B,H,Sqm,Skm,packed_q,packed_k,use_bias,p', [
    (5, 4, 100, 100, True, True, False, 0.0), (5, 4, 100, 100, True, True, True, 0.1), (4, 8, 100, 14, True, False, False, 0.0),
    (3, 2, 36, 50, True, True, True, 0.0), (6, 4, 14, 14, True, True, False, 0.0), (4, 4, 128, 128, True, True, False, 0.2)])
def test_mha_core_packed_rows(B, H, Sqm, Skm, packed_q, packed_k, use_bias, p):
    """Ragged batches without their padding rows (mmnas_mha_desc.q_off / k_off): sequence b owns the packed rows
    off[b] .. off[b+1]; lse / biasT / the dropout index keep the padded [B,H,...] layout.  Checked sequence by sequence
    against the fp64 reference of that sequence alone -- which is what the reference computes for the valid rows of a
    padded batch, its masked keys having probability exactly 0 (modules.py:195-196)."""
    import ctypes as C
    import mmnas_amd._lib as L
    from oracle import dropout_rng
    rs = np.random.RandomState(B * 7 + H + Sqm + Skm)
    dh, di = 64, H * 64
    lq = [int(rs.randint(1, Sqm + 1)) for _ in range(B)]
    lq[0], lq[-1] = Sqm, 1                       # the longest and the shortest
    self_att = packed_q and packed_k and Sqm == Skm
    lk = list(lq) if self_att else [int(rs.randint(1, Skm + 1)) for _ in range(B)]
    if not packed_q:
        lq = [Sqm] * B
    qoff = np.concatenate([[0], np.cumsum(lq)]).astype(np.int32)
    koff = np.concatenate([[0], np.cumsum(lk)]).astype(np.int32)
    Nq, Nk = int(qoff[-1]), (int(koff[-1]) if packed_k else B * Skm)
    Q, dO = rnd(rs, Nq, di), rnd(rs, Nq, di)
    K, V = rnd(rs, Nk, di), rnd(rs, Nk, di)
    mask = None
    if not packed_k:                             # padded keys with their mask (the guided operators: keys = the language stream)
        mask = np.zeros((B, Skm), np.bool_)
        for b in range(B):
            mask[b, lk[b]:] = True
    biasT = (rnd(rs, B, H, Skm, Sqm) * 2) if use_bias else None
    seed = 777
    Qd, Kd, Vd, dOd = g(Q), g(K), g(V), g(dO)
    bd = g(biasT) if use_bias else None
    O_ = torch.full((Nq, di), float('nan'), device=DEV)
    stats = torch.zeros(B, H, Sqm, 2, device=DEV)
    qo, ko = g(qoff), g(koff)
    m8 = g(mask.astype(np.uint8)) if mask is not None else None      # (kept alive: the descriptor holds a raw pointer)
    d = L.MhaDesc()
    d.B, d.H, d.Sq, d.Sk, d.dh = B, H, Sqm, Skm, dh
    d.ldq = d.ldk = d.ldv = d.ldo = di
    d.Q, d.K, d.V, d.mask, d.biasT, d.O, d.lse = (L.fptr(Qd), L.fptr(Kd), L.fptr(Vd), L.ptr(m8),
                                                 L.fptr(bd), L.fptr(O_), L.fptr(stats))
    d.q_off = L.ptr(qo) if packed_q else None
    d.k_off = L.ptr(ko) if packed_k else None
    d.drop_p, d.drop_site, d.drop_seed = p, 0, seed
    L.check(L.lib().mmnas_mha_core_fwd(C.byref(d), L.stream()))
    dQ, dK, dV = torch.full_like(Qd, float('nan')), torch.full_like(Kd, float('nan')), torch.full_like(Vd, float('nan'))
    dbT = torch.zeros(B, H, Skm, Sqm, device=DEV) if use_bias else None
    delta = torch.empty(B, H, Sqm, device=DEV)
    d.dO, d.dQ, d.dK, d.dV, d.dbiasT, d.delta = L.fptr(dOd), L.fptr(dQ), L.fptr(dK), L.fptr(dV), L.fptr(dbT), L.fptr(delta)
    L.check(L.lib().mmnas_mha_core_bwd(C.byref(d), L.stream()))
    torch.cuda.synchronize()
    dm_all = dropout_rng.scaled_mask(seed, 0, (B, H, Sqm, Skm), p) if p > 0 else None
    Oc, dQc, dKc, dVc = O_.cpu().numpy(), dQ.cpu().numpy(), dK.cpu().numpy(), dV.cpu().numpy()
    assert np.isfinite(Oc).all() and np.isfinite(dQc).all()
    for b in range(B):
        q0, q1 = int(qoff[b]), int(qoff[b + 1])
        k0, k1 = (int(koff[b]), int(koff[b + 1])) if packed_k else (b * Skm, b * Skm + lk[b])
        nq, nk = q1 - q0, k1 - k0
        Qt, Kt, Vt = (torch.from_numpy(a).double().unsqueeze(0).requires_grad_(True) for a in (Q[q0:q1], K[k0:k1], V[k0:k1]))
        bt = torch.from_numpy(biasT[b:b + 1, :, :nk, :nq].copy()).double().requires_grad_(True) if use_bias else None
        dm = torch.from_numpy(dm_all[b:b + 1, :, :nq, :nk].copy()).double() if p > 0 else None
        ref = _mha_ref(Qt, Kt, Vt, None, bt, H, dh, dm)
        assert rel_err(Oc[q0:q1], ref[0].detach().numpy()) < 1e-5, b
        ref.backward(torch.from_numpy(dO[q0:q1]).double().unsqueeze(0))
        assert rel_err(dQc[q0:q1], Qt.grad[0].numpy()) < 1e-4, b
        assert rel_err(dKc[k0:k1], Kt.grad[0].numpy()) < 1e-4, b
        assert rel_err(dVc[k0:k1], Vt.grad[0].numpy()) < 1e-4, b
        if use_bias:
            assert rel_err(dbT[b, :, :nk, :nq].cpu().numpy(), bt.grad[0].numpy()) < 1e-4, b
    if packed_k:       # a mask makes no sense beside packed keys: refused
        mz = g(np.zeros((B, Skm), np.uint8))
        d.mask = L.ptr(mz)
        with pytest.raises(L.MMNasHipError):
            L.check(L.lib().mmnas_mha_core_fwd(C.byref(d), L.stream()))


# ----------------------------------------------------------------------------- convs
@pytest.mark.parametrize('k', [3, 5, 7, 11])
def test_conv_building_blocks(k):
    from mmnas_amd import ops
    rs = np.random.RandomState(k)
    B, S, d = 3, 9, 64
    x = rnd(rs, B, S, d)
    w, b = rnd(rs, d, d, k) / math.sqrt(d * k), 0.1 * rnd(rs, d)
    wd, bd = rnd(rs, d, 1, k), 0.1 * rnd(rs, d)
    gy = rnd(rs, B, S, d)
    for dense in (True, False):
        xd = g(x).requires_grad_(True)
        W = g(w if dense else wd).requires_grad_(True)
        Bb = g(b if dense else bd).requires_grad_(True)
        y = ops.conv_seq(xd, W, Bb) if dense else ops.depthwise_conv_seq(xd, W, Bb)
        y.backward(g(gy))
        xt = torch.from_numpy(x).double().requires_grad_(True)
        Wt = torch.from_numpy(w if dense else wd).double().requires_grad_(True)
        Bt = torch.from_numpy(b if dense else bd).double().requires_grad_(True)
        r = torch.nn.functional.conv1d(xt.transpose(1, 2), Wt, Bt, padding=k // 2, groups=1 if dense else d).transpose(1, 2)
        r.backward(torch.from_numpy(gy).double())
        assert rel_err(y.detach().cpu().numpy(), r.detach().numpy()) < 1e-5
        assert rel_err(xd.grad.cpu().numpy(), xt.grad.numpy()) < 1e-4
        assert rel_err(W.grad.cpu().numpy(), Wt.grad.numpy()) < 1e-4
        assert rel_err(Bb.grad.cpu().numpy(), Bt.grad.numpy()) < 1e-4


def test_pack_adam_sumsq():
    import ctypes as C
    import mmnas_amd._lib as L
    rs = np.random.RandomState(1)
    n = 10007
    p0, g0 = rnd(rs, n), rnd(rs, n) * 3
    p, gr = g(p0), g(g0)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    ss = torch.zeros(1, device=DEV)
    L.check(L.lib().mmnas_sumsq(L.fptr(gr), n, L.fptr(ss), L.stream()))
    assert abs(float(ss) - float((g0.astype(np.float64) ** 2).sum())) < 1e-3 * float(ss)
    pt = torch.from_numpy(p0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    for step in (1, 2, 3):
        L.check(L.lib().mmnas_adam_step(L.fptr(p), L.fptr(gr), L.fptr(m), L.fptr(v), n, 1e-3, 0.9, 0.98, 1e-9, 0.0,
                                        L.fptr(ss), 1.0, step, L.stream()))
        pt.grad = torch.from_numpy(g0.copy())
        torch.nn.utils.clip_grad_norm_([pt], 1.0)
        opt.step()
    assert rel_err(p.cpu().numpy(), pt.detach().numpy()) < 1e-5
    # pack / unpack
    a, b = g(rnd(rs, 1000)), g(rnd(rs, 333))
    segs = (L.Segment * 2)()
    segs[0].ptr, segs[0].offset, segs[0].n = a.data_ptr(), 0, 1000
    segs[1].ptr, segs[1].offset, segs[1].n = b.data_ptr(), 1000, 333
    sd = torch.frombuffer(bytearray(bytes(segs)), dtype=torch.uint8).to(DEV)
    stg = torch.zeros(1333, device=DEV)
    L.check(L.lib().mmnas_pack_segments(sd.data_ptr(), 2, L.fptr(stg), 0.5, 0, L.stream()))
    assert torch.allclose(stg[:1000], a * 0.5) and torch.allclose(stg[1000:], b * 0.5)
    a0, b0 = a.clone(), b.clone()
    L.check(L.lib().mmnas_pack_segments(sd.data_ptr(), 2, L.fptr(stg), 2.0, 1, L.stream()))
    assert torch.allclose(a, a0) and torch.allclose(b, b0)
    # the host-table variant (table in the kernel arguments, chunks of 96 records): 200 ragged segments
    sizes = [int(x) for x in rs.randint(1, 700, size=200)]
    flat = g(rnd(rs, sum(sizes) + 64 * 200))
    many = (L.Segment * 200)()
    off_src = off_stg = 0
    for i, nseg in enumerate(sizes):
        many[i].ptr, many[i].offset, many[i].n = flat.data_ptr() + 4 * off_src, off_stg, nseg
        off_src += nseg + 64
        off_stg += nseg
    stg2 = torch.zeros(off_stg, device=DEV)
    L.check(L.lib().mmnas_pack_segments_host(many, 200, L.fptr(stg2), 1.0, 0, L.stream()))
    exp = torch.cat([flat[o:o + nseg] for o, nseg in zip(np.cumsum([0] + [x + 64 for x in sizes[:-1]]), sizes)])
    assert torch.equal(stg2, exp)
    before = flat.clone()
    L.check(L.lib().mmnas_pack_segments_host(many, 200, L.fptr(stg2 * 3.0), 1.0, 1, L.stream()))
    o = 0
    for nseg in sizes:
        assert torch.equal(flat[o:o + nseg], before[o:o + nseg] * 3.0)
        assert torch.equal(flat[o + nseg:o + nseg + 64], before[o + nseg:o + nseg + 64])     # gaps untouched
        o += nseg + 64


@pytest.mark.parametrize('B,T,E,H', [(64, 14, 300, 512), (3, 5, 20, 64), (70, 3, 300, 256), (64, 14, 300, 256),
                                     (160, 50, 300, 512), (2, 1, 24, 128), (130, 7, 28, 64)])
def test_lstm_vs_torch_fp64(B, T, E, H):
    """Persistent-kernel LSTM (mmnas_lstm_seq_fwd/bwd: one launch per pass, the state handed between workgroups once
    per step) against torch.nn.LSTM evaluated in float64 on the CPU: output sequence and every gradient (input, both
    weight matrices, both biases).  Several sample blocks (B > 64, ragged last block), one step, 50 steps; repeated
    to catch a hand-off that only fails now and then."""
    from mmnas_amd import ops, _lib as L
    torch.manual_seed(B + T + H)
    ref = torch.nn.LSTM(input_size=E, hidden_size=H, num_layers=1, batch_first=True).double()
    x = torch.randn(B, T, E, dtype=torch.float64, requires_grad=True)
    go = torch.randn(B, T, H, dtype=torch.float64)
    y = ref(x)[0]
    y.backward(go)
    mod = torch.nn.LSTM(input_size=E, hidden_size=H, num_layers=1, batch_first=True)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.to(DEV)
    xd = x.detach().float().to(DEV).requires_grad_(True)
    assert ops.lstm_supported(xd, mod)
    for rep in range(4):
        xd.grad = None
        mod.zero_grad()
        junk = torch.randn(1 << 22, device=DEV).mul_(2.0)        # other work in flight, caches disturbed
        yd = ops.lstm(xd, mod)
        yd.backward(go.float().to(DEV))
        assert L.lib().mmnas_lstm_seq_timed_out(L.stream()) == 0
        assert rel_err(yd.detach().cpu().numpy(), y.detach().numpy()) < 2e-5, rep
        assert rel_err(xd.grad.cpu().numpy(), x.grad.numpy()) < 1e-4, rep
        for k, p in mod.named_parameters():
            want = dict(ref.named_parameters())[k].grad.numpy()
            assert rel_err(p.grad.cpu().numpy(), want) < 1e-4, (k, rep)
        del junk


def test_embedding_backward_adds_rows_into_the_gradient():
    """ops.embedding: forward = nn.Embedding; backward adds dy rows into the weight gradient (duplicates accumulate),
    with and without a gradient sink (flat gradient buffer)."""
    from mmnas_amd import ops, dp
    torch.manual_seed(3)
    V, E, B, S = 50, 300, 7, 14
    ref = torch.nn.Embedding(V, E).double()
    idx = torch.randint(0, V, (B, S))
    idx[:, -3:] = 0                       # many duplicates of token 0 (the padding token of the loaders)
    go = torch.randn(B, S, E, dtype=torch.float64)
    ref(idx).backward(go)
    for use_sink in (False, True):
        mod = torch.nn.Embedding(V, E)
        mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        mod = mod.to(DEV)
        if use_sink:
            fg = dp.FlatGrads([mod.weight])
            fg.attach()
            fg.enable_sinks(None)
        y = ops.embedding(idx.to(DEV), mod)
        assert torch.equal(y.detach().cpu(), mod.weight.detach().cpu()[idx])
        y.backward(go.float().to(DEV))
        assert rel_err(mod.weight.grad.cpu().numpy(), ref.weight.grad.numpy()) < 1e-6
        if use_sink:
            assert mod.weight.grad.data_ptr() == fg.views[0].data_ptr()


def test_embedding_backward_fixed_order_for_the_row_exchange():
    """mmnas_embedding_bwd_det (dp.RowExchange applies every rank's (token index, dy row) pairs with it): equals the
    scatter-add in float64, scales, ignores indices outside the table, accumulates onto what the gradient already holds,
    and is bitwise reproducible (no atomics: the ranks' tables must stay identical)."""
    from mmnas_amd import _lib as L
    rs = np.random.RandomState(11)
    for V, E, n in ((50, 300, 2000), (20000, 300, 7168), (7, 24, 3), (300, 1024, 700)):
        idx = rs.randint(-2, V + 2, size=n).astype(np.int64)          # a few out-of-range entries
        idx[::2] = idx[0] if 0 <= idx[0] < V else 1                   # one row hit by half of the tokens (the padding token)
        dy = rs.randn(n, E).astype(np.float32)
        base = rs.randn(V, E).astype(np.float32)
        want = base.astype(np.float64)
        ok = (idx >= 0) & (idx < V)
        np.add.at(want, idx[ok], 0.25 * dy[ok].astype(np.float64))
        outs = []
        idx_d, dy_d = g(idx), g(dy)
        for rep in range(3):
            dW = g(base.copy())
            junk = torch.randn(1 << 20, device=DEV)
            ws = torch.full((L.lib().mmnas_embedding_bwd_det_ws_floats(n, E),), float('nan'), device=DEV)
            L.check(L.lib().mmnas_embedding_bwd_det(L.ptr(idx_d), L.fptr(dy_d), L.fptr(dW), L.fptr(ws), n, E, V, 0.25, L.stream()))
            outs.append(dW.cpu().numpy())
            del junk
        assert rel_err(outs[0], want) < 2e-6, (V, E, n)
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_pack_segments_balanced_over_very_uneven_segments():
    """The gather / scatter of a gradient bucket: one 24 MB segment (the embedding table of the stem bucket) beside small
    and unaligned ones -- every float moves exactly once in both directions, gaps stay untouched."""
    from mmnas_amd import _lib as L
    rs = np.random.RandomState(5)
    sizes = [6_000_000, 3, 8191, 8192, 8193, 1_048_576, 1, 70_001]
    gaps = [5, 1, 0, 3, 64, 2, 7, 0]
    flat = g(rnd(rs, sum(sizes) + sum(gaps) + 8))
    segs = (L.Segment * len(sizes))()
    o = so = 0
    src_off = []
    for i, (n, gp) in enumerate(zip(sizes, gaps)):
        segs[i].ptr, segs[i].offset, segs[i].n = flat.data_ptr() + 4 * o, so, n
        src_off.append(o)
        o += n + gp
        so += n
    stg = torch.full((so,), float('nan'), device=DEV)
    L.check(L.lib().mmnas_pack_segments_host(segs, len(sizes), L.fptr(stg), 0.5, 0, L.stream()))
    exp = torch.cat([flat[a:a + n] for a, n in zip(src_off, sizes)]) * 0.5
    assert torch.equal(stg, exp)
    before = flat.clone()
    L.check(L.lib().mmnas_pack_segments_host(segs, len(sizes), L.fptr(stg), 4.0, 1, L.stream()))
    for a, n, gp in zip(src_off, sizes, gaps):
        assert torch.equal(flat[a:a + n], before[a:a + n] * 2.0)
        assert torch.equal(flat[a + n:a + n + gp], before[a + n:a + n + gp])


@pytest.mark.parametrize('mode', [6, 0])
def test_gemm_nine_groups_with_their_own_dropout_seeds(mode, gemm_tuning):
    """One launch for every projection of a supernet node's attention candidates (mixed chains): up to 9 groups with
    different row counts, operands and outputs; and the merge projections of the candidates as groups that share the
    launch's dropout rate and site but draw from their OWN seeds (the reference's modules own their dropout)."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    gemm_tuning(split=mode)
    rs = np.random.RandomState(91)
    N, K = 256, 256
    Ms = [6400, 6400, 6400, 6400, 6400, 6400, 6400, 896, 896]
    A = [g(rnd(rs, M, K)) for M in Ms]
    B = [g(rnd(rs, N, K)) for _ in Ms]
    Cs = [torch.full((M, N), float('nan'), device=DEV) for M in Ms]
    ops.gemm(L.GEMM_NT, [dict(M=M, A=[a], B=[b], C=c) for M, a, b, c in zip(Ms, A, B, Cs)], N, K, K, K, N)
    for a, b, c in zip(A, B, Cs):
        ref = a.double() @ b.double().t()
        assert rel_err(c.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    # three merge groups: bias-free product, dropout (site 1, p = 0.1) with a seed per group, then the residual
    p, seeds = 0.1, [0x1234567811, 0xABCDEF0122, 0x5555AAAA33]
    res = g(rnd(rs, 6400, N))
    Cm = [torch.empty(6400, N, device=DEV) for _ in seeds]
    desc = ops.gemm_desc(L.GEMM_NT, [dict(M=6400, A=[A[j]], B=[B[j]], C=Cm[j], residual=res) for j in range(3)], N, K, K, K, N,
                         drop=(p, seeds[0], 1), ldres=N)
    for j in range(3):
        desc.g[j].drop_seed = seeds[j]
    import ctypes as C
    L.check(L.lib().mmnas_gemm(C.byref(desc), L.stream()))
    masks = []
    for j in range(3):
        mask = ops.dropout_mask(6400 * N, p, seeds[j], 1, DEV).view(6400, N)
        masks.append(mask)
        ref = (A[j].double() @ B[j].double().t()) * mask.double() + res.double()
        assert rel_err(Cm[j].cpu().numpy(), ref.cpu().numpy()) < 1e-5, j
    assert not torch.equal(masks[0], masks[1]) and not torch.equal(masks[1], masks[2])


@pytest.mark.parametrize('layout', ['NT', 'NN'])
@pytest.mark.parametrize('ngroups', [1, 2, 5, 9])
def test_gemm_group_of_a_tile_with_ragged_row_counts(layout, ngroups, gemm_tuning):
    """The group a tile belongs to is found from the header's copy of the group boundaries (a count of boundaries at or
    below the tile, INT_MAX past the last group): row counts of one row, just under / at / just over the 64- and 128-row
    tile edges, and every number of groups up to the nine the descriptor holds."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    gemm_tuning(split=6)
    rs = np.random.RandomState(17 + ngroups)
    N, K = 192, 96
    Ms = [1, 63, 64, 65, 127, 128, 129, 896, 6400][:ngroups] if ngroups < 9 else [129, 1, 6400, 63, 64, 896, 65, 127, 128]
    groups, refs = [], []
    for M in Ms:
        a, c = g(rnd(rs, M, K)), torch.full((M, N), float('nan'), device=DEV)
        b = g(rnd(rs, N, K)) if layout == 'NT' else g(rnd(rs, K, N))
        bias = g(rnd(rs, N))
        groups.append(dict(M=M, A=[a], B=[b], C=c, bias=bias))
        refs.append(a.double() @ (b.double().t() if layout == 'NT' else b.double()) + bias.double())
    if layout == 'NT':
        ops.gemm(L.GEMM_NT, groups, N, K, K, K, N)
    else:
        ops.gemm(L.GEMM_NN, groups, N, K, K, N, N)
    for grp, ref in zip(groups, refs):
        assert rel_err(grp['C'].cpu().numpy(), ref.cpu().numpy()) < 1e-5, grp['M']


@pytest.mark.parametrize('layout', ['NT', 'NN'])
@pytest.mark.parametrize('wgs', [7, 33, 129])
def test_gemm_streamed_pieces_across_k_segments(layout, wgs, gemm_tuning):
    """Stream-K pieces that start in the middle of a K-segment and run across segment boundaries: the segment / K-tile of
    a piece's loads is carried from its first unit (one division per piece), not divided out per tile."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    gemm_tuning(split=6, sk=2, min_units=1, wgs=wgs)
    rs = np.random.RandomState(300 + wgs)
    M, N, K = 320, 192, 160          # 5 K-tiles per segment, 3 segments: 15 units per tile, 15 tiles
    As = [g(rnd(rs, M, K)) for _ in range(3)]
    Bs = [g(rnd(rs, N, K)) if layout == 'NT' else g(rnd(rs, K, N)) for _ in range(3)]
    C = torch.full((M, N), float('nan'), device=DEV)
    if layout == 'NT':
        ops.gemm(L.GEMM_NT, [dict(M=M, A=As, B=Bs, C=C)], N, K, K, K, N, nseg=3)
        ref = sum(a.double() @ b.double().t() for a, b in zip(As, Bs))
    else:
        ops.gemm(L.GEMM_NN, [dict(M=M, A=As, B=Bs, C=C)], N, K, K, N, N, nseg=3)
        ref = sum(a.double() @ b.double() for a, b in zip(As, Bs))
    assert rel_err(C.cpu().numpy(), ref.cpu().numpy()) < 1e-5


@pytest.mark.parametrize('rows,K,bias', [(6400, 512, True), (37, 512, False), (896, 1024, True), (160, 300, True), (5, 4, True)])
def test_one_unit_linear_matrix_vector_kernels(rows, K, bias):
    """mmnas_glimpse1_fwd / _bwd: nn.Linear with one output unit (AttFlat's glimpse logits, the ITM matching score) as a
    matrix-vector product and, backward, one pass over x with the column sums reduced by a second launch -- against float64;
    dw / db are ADDED to their buffers; and through ops.linear, which routes N = 1 here."""
    import ctypes as C
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    gen = torch.Generator(device=DEV).manual_seed(9)
    r = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    x, w, b, dy = r(rows, K), r(K), r(1) if bias else None, r(rows)
    lib = L.lib()
    assert lib.mmnas_glimpse1_supported(K) == 1 and lib.mmnas_glimpse1_supported(2048) == 0 and lib.mmnas_glimpse1_supported(6) == 0
    y = torch.empty(rows, device=DEV)
    L.check(lib.mmnas_glimpse1_fwd(L.fptr(x), L.fptr(w), L.fptr(b), L.fptr(y), rows, K, L.stream()))
    ref = x.double() @ w.double() + (b.double() if bias else 0.0)
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    dx, dw0, db0 = torch.empty_like(x), r(K), r(1)
    dw, db = dw0.clone(), db0.clone()
    ws = torch.empty(lib.mmnas_glimpse1_bwd_ws_floats(rows, K), device=DEV)
    L.check(lib.mmnas_glimpse1_bwd(L.fptr(dy), L.fptr(x), L.fptr(w), L.fptr(dx), L.fptr(dw), L.fptr(db if bias else None), L.fptr(ws),
                                   rows, K, L.stream()))
    assert torch.equal(dx, dy[:, None] * w[None, :])
    rw = dw0.double() + dy.double() @ x.double()
    assert float((dw.double() - rw).abs().max()) <= 2e-6 * float(rw.abs().max())
    if bias:
        assert abs(float(db.double() - db0.double() - dy.double().sum())) <= 2e-6 * float(dy.abs().sum())
    # through the operator layer (autograd): the same numbers
    xa, wa = x.clone().requires_grad_(True), w.clone()[None, :].requires_grad_(True)
    ba = b.clone().requires_grad_(True) if bias else None
    ya = ops.linear(xa, wa, ba)
    assert ya.shape == (rows, 1) and torch.equal(ya[:, 0], y)
    ya.backward(dy[:, None])
    assert torch.equal(xa.grad, dx)
    assert float((wa.grad[0].double() - dy.double() @ x.double()).abs().max()) <= 2e-6 * float(rw.abs().max()) + 1e-6


@pytest.mark.parametrize('layout,Ms,N,K,nseg,epi', [
    ('NT', [6400], 256, 256, 1, ''), ('NT', [300, 77, 130], 256, 64, 1, 'bRd'), ('NT', [100], 72, 96, 1, 'bdr'),
    ('NT', [6397], 256, 256, 1, 'dr'), ('NN', [333], 128, 160, 1, 'g'), ('NN', [6400], 1024, 256, 1, 'gc'),
    ('NN', [333, 70], 72, 160, 1, 'gcr'),
    ('NN', [500], 256, 64, 3, 'r'), ('NN', [6400], 256, 256, 1, 'a'), ('NT', [129], 8, 32, 1, 'b')])
def test_gemm_lean_kernels_equal_the_general_kernel(layout, Ms, N, K, nseg, epi, monkeypatch):
    """Whole-tile NT / NN products run on the lean kernels (short set-up, transposed accumulators, 16-byte epilogue rows;
    gemm_body<..., LEAN>): the same K loop and the same arithmetic per element, so results equal the general kernel's bit
    for bit -- every epilogue term (bias, relu, dropout, gate, residual, accumulate), ragged M and N, grouped problems,
    K-segments.  Column sums (the element-wise lean form of NN products) are float atomics: checked to round-off."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    gen = torch.Generator(device=DEV).manual_seed(5)
    r = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    data = []
    for M in Ms:
        d = dict(A=[r(M, K) for _ in range(nseg)], B=[r(N, K) if layout == 'NT' else r(K, N) for _ in range(nseg)],
                 bias=r(N) if 'b' in epi else None, residual=r(M, N) if 'r' in epi else None,
                 gate=r(M, N) if 'g' in epi else None, C0=r(M, N))
        data.append(d)
    kw = dict(nseg=nseg, relu='R' in epi, accumulate='a' in epi)
    if 'd' in epi:
        kw['drop'] = (0.2, 4242, 5)
    if 'g' in epi:
        kw.update(gate_scale=1.25, ldgate=N)
    if 'r' in epi:
        kw['ldres'] = N
    outs = {}
    for lean in ('0', '3'):
        monkeypatch.setenv('MMNAS_GEMM_LEAN', lean)
        L.check(L.lib().mmnas_gemm_reload_tuning())
        groups = [dict(M=M, A=d['A'], B=d['B'], C=d['C0'].clone(), bias=d['bias'], residual=d['residual'], gate=d['gate'],
                       colsum=torch.zeros(N, device=DEV) if 'c' in epi else None) for M, d in zip(Ms, data)]
        ops.gemm(L.GEMM_NT if layout == 'NT' else L.GEMM_NN, groups, N, K, K, K if layout == 'NT' else N, N, **kw)
        torch.cuda.synchronize()
        outs[lean] = groups
    monkeypatch.delenv('MMNAS_GEMM_LEAN')
    L.check(L.lib().mmnas_gemm_reload_tuning())
    for a, b, d, M in zip(outs['0'], outs['3'], data, Ms):
        assert torch.equal(a['C'], b['C'])
        if 'c' in epi:
            ref = a['C'].double().sum(0)
            assert float((b['colsum'].double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-6
            assert float((a['colsum'].double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-6
    # and against float64 (no epilogue terms beyond bias / residual: the plain cases)
    if epi in ('', 'r', 'b'):
        for b, d in zip(outs['3'], data):
            ref = sum(x.double() @ (w.double().t() if layout == 'NT' else w.double()) for x, w in zip(d['A'], d['B']))
            if d['bias'] is not None:
                ref = ref + d['bias'].double()
            if d['residual'] is not None:
                ref = ref + d['residual'].double()
            assert float((b['C'].double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize('Ms,N,K', [([256], 256, 6400), ([256, 256, 256], 256, 1600), ([96], 72, 1888), ([1024], 256, 3517), ([64], 64, 1888)])
def test_gemm_lean_weight_gradient_pieces(Ms, N, K, monkeypatch):
    """Split-K weight gradients (TN, "C +=") on the lean kernel: pieces found by multiply-shift, added by buffer atomics with
    the rows behind M dropped by the range check and the columns behind N masked; one tile (no split) stays on the general
    kernel."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    gen = torch.Generator(device=DEV).manual_seed(6)
    r = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    data = [dict(A=r(K, M), B=r(K, N), C0=r(M, N)) for M in Ms]
    for lean in ('0', '3'):
        monkeypatch.setenv('MMNAS_GEMM_LEAN', lean)
        L.check(L.lib().mmnas_gemm_reload_tuning())
        groups = [dict(M=M, A=[d['A']], B=[d['B']], C=d['C0'].clone()) for M, d in zip(Ms, data)]
        if len(set(Ms)) == 1:
            ops.gemm(L.GEMM_TN, groups, N, K, Ms[0], N, N, accumulate=True)
        torch.cuda.synchronize()
        for g, d in zip(groups, data):
            ref = d['C0'].double() + d['A'].double().t() @ d['B'].double()
            assert float((g['C'].double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    monkeypatch.delenv('MMNAS_GEMM_LEAN')
    L.check(L.lib().mmnas_gemm_reload_tuning())


@pytest.mark.parametrize('Ms,N,K', [([200], 64, 32), ([6400], 256, 256), ([896, 77, 6400], 128, 512), ([130], 512, 2048)])
def test_gemm_weight_planes_by_lds_dma_equal_the_in_kernel_split(Ms, N, K):
    """mmnas_gemm_desc.b_planes: the weight operand as the three bf16 planes of mmnas_split_planes, streamed global -> LDS by
    LDS-DMA -- the same split, the same MFMA order: the product on the planes equals the product on the fp32 matrix bit
    for bit (ragged row counts, several groups, bias + ReLU epilogue), and the planes sum back to the matrix exactly."""
    import ctypes as C
    from mmnas_amd import _lib as L, ops
    g = torch.Generator().manual_seed(7)
    As = [torch.randn(M, K, generator=g).cuda() for M in Ms]
    Ws = [(torch.randn(N, K, generator=g) * 0.1).cuda() for _ in Ms]
    Ps = [ops.split_planes(W) for W in Ws]
    for W, P in zip(Ws, Ps):
        assert P.dtype == torch.bfloat16 and tuple(P.shape) == (3, N, K)
        assert torch.equal(P[0].double() + P[1].double() + P[2].double(), W.double())
    bias = torch.randn(N, generator=g).cuda()
    outs = []
    for planes in (False, True):
        Cs = [torch.full((M, N), float('nan'), device='cuda') for M in Ms]
        d = ops.gemm_desc(L.GEMM_NT, [dict(M=M, A=[a], B=[(p if planes else w)], C=c, bias=bias)
                                      for M, a, w, p, c in zip(Ms, As, Ws, Ps, Cs)], N, K, K, K, N, relu=True, b_planes=planes)
        L.check(L.lib().mmnas_gemm(C.byref(d), L.stream()))
        torch.cuda.synchronize()
        outs.append(Cs)
    for c0, c1, a, w in zip(outs[0], outs[1], As, Ws):
        assert torch.equal(c0, c1)
        ref = torch.relu(a.double() @ w.double().t() + bias.double())
        assert float((c1.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-6
    # shapes the LDS-DMA kernel does not cover are refused, not misread
    with pytest.raises(L.MMNasHipError):
        d = ops.gemm_desc(L.GEMM_NN, [dict(M=Ms[0], A=[As[0]], B=[Ps[0]], C=outs[0][0])], N, K, K, N, N, b_planes=True)
        L.check(L.lib().mmnas_gemm(C.byref(d), L.stream()))


@pytest.mark.parametrize('B,S,d,k', [(3, 9, 64, 3), (2, 14, 128, 11), (5, 100, 64, 7), (4, 7, 96, 5)])
def test_conv_seq_overlapping_rows_vs_conv1d_fp64(B, S, d, k, monkeypatch):
    """StdConv's core (modules.py:472,480-481: nn.Conv1d over the sequence, zero 'same' padding) as ONE product on the
    zero-padded input read with overlapping rows (lda = d, K = k d) -- no im2col buffer -- against torch's conv1d in fp64,
    forward and all three gradients; and against the explicit-window form it replaces."""
    from mmnas_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, S, d, generator=g)
    w = torch.randn(d, d, k, generator=g) * 0.1
    b = torch.randn(d, generator=g)
    dy = torch.randn(B, S, d, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.conv1d(xr.transpose(1, 2), wr, br, padding=k // 2).transpose(1, 2)
    yr.backward(dy.double())
    outs = {}
    for mode in ('direct', 'im2col'):
        monkeypatch.setenv('MMNAS_CONV_IM2COL', '1' if mode == 'im2col' else '0')
        xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
        calls = []
        orig = ops.ConvSeqFn.apply
        monkeypatch.setattr(ops.ConvSeqFn, 'apply', lambda *a: (calls.append(1), orig(*a))[1])
        y = ops.conv_seq(xg, wg, bg)
        monkeypatch.setattr(ops.ConvSeqFn, 'apply', orig)
        assert len(calls) == (1 if mode == 'direct' else 0)
        y.backward(dy.cuda())
        torch.cuda.synchronize()
        outs[mode] = (y.detach(), xg.grad, wg.grad, bg.grad)
        for got, ref in zip(outs[mode], (yr, xr.grad, wr.grad, br.grad)):
            assert got.shape == ref.shape
            assert float((got.detach().double().cpu() - ref.detach()).abs().max()) <= 3e-6 * float(ref.abs().max()), mode
    for a, c in zip(outs['direct'], outs['im2col']):
        assert float((a - c).abs().max()) <= 3e-6 * float(c.abs().max())
    # the re-arranged copies of the weight are re-made when (and only when) the parameter has been written
    wg = w.cuda().requires_grad_(True)
    f0, r0 = ops._conv_weights(wg)
    assert ops._conv_weights(wg)[0] is f0
    with torch.no_grad():
        wg.mul_(2.0)
    f1, r1 = ops._conv_weights(wg)
    assert f1 is not f0 and torch.equal(f1.view(d, k, d), wg.detach().permute(0, 2, 1))
    assert torch.equal(r1.view(k, d, d), wg.detach().flip(2).permute(2, 0, 1))
    # mmnas_pad_seq: the padded row grid
    xp = ops._pad_seq(x.cuda(), k // 2, S + 2 * (k // 2), k)
    ref = torch.nn.functional.pad(x, (0, 0, k // 2, k // 2)).reshape(-1, d)
    assert xp.shape[0] % 32 == 0 and xp.shape[0] >= ref.shape[0] + k
    assert torch.equal(xp[:ref.shape[0]].cpu(), ref) and not bool(xp[ref.shape[0]:].any())


@pytest.mark.parametrize('M,N,K', [(256, 256, 3517), (1024, 256, 77), (64, 128, 33), (512, 512, 6401)])
@pytest.mark.parametrize('accumulate', [False, True])
def test_gemm_tn_ragged_reduction_length(M, N, K, accumulate):
    """Weight-gradient products (TN) whose reduction length -- the row count of a PACKED ragged batch -- is no multiple of
    the 32-deep K-tile: they stay on the buffer-load kernels (the rows behind K lie outside the operands' ranges and read
    as zero) instead of dropping to the guarded-load path, with the same result."""
    from mmnas_amd import ops
    import mmnas_amd._lib as L
    rs = np.random.RandomState(M + N + K)
    A, B = rnd(rs, K, M), rnd(rs, K, N)
    C0 = rnd(rs, M, N)
    ref = torch.from_numpy(A).double().t() @ torch.from_numpy(B).double() + (torch.from_numpy(C0).double() if accumulate else 0)
    C = g(C0.copy()) if accumulate else torch.full((M, N), float('nan'), device=DEV)
    L.check(L.lib().mmnas_prof_enable(1))
    try:
        ops.gemm(L.GEMM_TN, [dict(M=M, A=[g(A)], B=[g(B)], C=C)], N, K, M, N, N, accumulate=accumulate)
        torch.cuda.synchronize()
    finally:
        arr = (L.ProfStat * len(L.K_NAMES))()
        L.check(L.lib().mmnas_prof_collect(arr))
        L.check(L.lib().mmnas_prof_enable(0))
    assert rel_err(C.cpu().numpy(), ref.numpy()) < 3e-6
