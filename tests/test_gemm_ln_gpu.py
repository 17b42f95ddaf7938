"""mmnas_gemm_ln (gemmln.hip): the merge / last feed-forward projection with its dropout + residual epilogue and the
LayerNorm behind it as ONE launch, against (1) a float64 torch restatement of `norm(x + dropout(linear(att)))`
(modules.py:44-56, 186-187, 261-271; dropout by mask replay: kernels and the numpy restatement derive the same keep mask
from (seed, site, index)) and (2) the two-launch form mmnas_gemm + mmnas_layernorm_fwd it replaces.  Tolerance 1e-3 relative
(BASELINE north_star) on z and y; measured ~2e-6."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests.util import TOL, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'
os.environ.setdefault('MMNAS_GEMM_LN_MINM', '0'); os.environ.setdefault('MMNAS_GEMM_LN_MAXK', '65536')   # (read once by the library: the small row counts below run the panel kernel too)


def _case(rs, M, K, bias, residual, ragged_ld):
    N = 256
    lda = K + (8 if ragged_ld else 0)
    A = torch.from_numpy(rs.standard_normal((M, lda)).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(DEV) if bias else None
    R = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32)).to(DEV) if residual else None
    la = torch.from_numpy((1.0 + 0.3 * rs.standard_normal(N)).astype(np.float32)).to(DEV)
    lb = torch.from_numpy((0.3 * rs.standard_normal(N)).astype(np.float32)).to(DEV)
    return N, lda, A, W, b, R, la, lb


def _run(L, ops, M, K, N, lda, A, W, b, R, la, lb, drop, panel, want_z=True, planes=False):
    z = torch.full((M, N), float('nan'), device=DEV) if want_z else None
    y = torch.full((M, N), float('nan'), device=DEV)
    Wop = ops.split_planes(W) if planes else W      # [3, N, K] bf16: W = p0 + p1 + p2 exactly
    d = ops.gemm_desc(L.GEMM_NT, [dict(M=M, A=[A], B=[Wop], C=z if want_z else y, bias=b, residual=R)], N, K, lda, K, N,
                      drop=drop, ldres=N if R is not None else 0, b_planes=planes)
    if not want_z:
        d.g[0].C = None
    old = L.lib().mmnas_set_gemm_ln(1 if panel else 0)
    try:
        L.check(L.lib().mmnas_gemm_ln(C.byref(d), L.fptr(la), L.fptr(lb), L.fptr(y), 1e-6, L.stream()))
        torch.cuda.synchronize()
    finally:
        L.lib().mmnas_set_gemm_ln(old)
    return z, y


def _ref64(A, W, b, R, la, lb, K, drop):
    from oracle import dropout_rng
    t = A[:, :K].double().cpu() @ W.double().cpu().t()
    if b is not None:
        t = t + b.double().cpu()
    if drop is not None:
        p, seed, site = drop
        t = t * torch.from_numpy(dropout_rng.scaled_mask(seed, site, tuple(t.shape), p).astype(np.float64))
    if R is not None:
        t = t + R.double().cpu()
    mean = t.mean(-1, keepdim=True)
    std = t.std(-1, keepdim=True)                # Bessel-corrected, eps on the std (modules.py:52-56)
    return t, la.double().cpu() * (t - mean) / (std + 1e-6) + lb.double().cpu()


@pytest.mark.parametrize('M,K', [(6400, 256), (6400, 1024), (896, 256), (100, 64), (33, 128), (1, 256), (4097, 192)])
@pytest.mark.parametrize('bias,residual,dropout', [(True, True, True), (False, True, False), (True, False, True), (False, False, False)])
def test_gemm_ln_panel_vs_float64_and_the_two_launch_form(M, K, bias, residual, dropout):
    from mmnas_amd import _lib as L, ops
    rs = np.random.RandomState(M * 7 + K)
    N, lda, A, W, b, R, la, lb = _case(rs, M, K, bias, residual, ragged_ld=(K == 192))
    drop = (0.1, 0x1234567887654321 + M, 1) if dropout else None
    z, y = _run(L, ops, M, K, N, lda, A, W, b, R, la, lb, drop, panel=True)
    z2, y2 = _run(L, ops, M, K, N, lda, A, W, b, R, la, lb, drop, panel=False)
    zr, yr = _ref64(A, W, b, R, la, lb, K, drop)
    assert torch.isfinite(z).all() and torch.isfinite(y).all()
    assert rel_err(z.cpu().numpy(), zr.numpy()) < TOL and rel_err(y.cpu().numpy(), yr.numpy()) < TOL
    # the two forms compute the same products on the same split operands: z agrees to accumulation order, y likewise
    assert rel_err(z.cpu().numpy(), z2.cpu().numpy()) < 2e-5 and rel_err(y.cpu().numpy(), y2.cpu().numpy()) < 2e-5
    assert rel_err(y.cpu().numpy(), yr.numpy()) < 2e-5      # measured ~2e-6: fp32-grade products, fp32 statistics
    # the weight operand as pre-split bf16 planes streamed by LDS-DMA: the same split, the same products -- bit for bit
    z3, y3 = _run(L, ops, M, K, N, lda, A, W, b, R, la, lb, drop, panel=True, planes=True)
    assert torch.equal(z3, z) and torch.equal(y3, y)


def test_gemm_ln_without_z_and_rows_behind_m_untouched():
    """z may be NULL (nobody needs the pre-LayerNorm sum); rows behind M of a larger y allocation are not written."""
    from mmnas_amd import _lib as L, ops
    rs = np.random.RandomState(5)
    M, K = 70, 128
    N, lda, A, W, b, R, la, lb = _case(rs, M, K, True, True, False)
    ybig = torch.full((M + 26, N), 7.0, device=DEV)
    d = ops.gemm_desc(L.GEMM_NT, [dict(M=M, A=[A], B=[W], C=ybig, bias=b, residual=R)], N, K, lda, K, N, ldres=N)
    d.g[0].C = None
    old = L.lib().mmnas_set_gemm_ln(1)       # (the panel kernel is opt-in; the two-launch form needs z)
    try:
        L.check(L.lib().mmnas_gemm_ln(C.byref(d), L.fptr(la), L.fptr(lb), L.fptr(ybig), 1e-6, L.stream()))
        torch.cuda.synchronize()
    finally:
        L.lib().mmnas_set_gemm_ln(old)
    _, yr = _ref64(A, W, b, R, la, lb, K, None)
    assert rel_err(ybig[:M].cpu().numpy(), yr.numpy()) < 2e-5
    assert bool((ybig[M:] == 7.0).all())


def test_gemm_ln_other_widths_take_the_two_launch_form():
    from mmnas_amd import _lib as L, ops
    rs = np.random.RandomState(9)
    M, K, N = 200, 128, 512
    A = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(DEV)
    la, lb = torch.ones(N, device=DEV), torch.zeros(N, device=DEV)
    z, y = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    d = ops.gemm_desc(L.GEMM_NT, [dict(M=M, A=[A], B=[W], C=z)], N, K, K, K, N)
    L.check(L.lib().mmnas_gemm_ln(C.byref(d), L.fptr(la), L.fptr(lb), L.fptr(y), 1e-6, L.stream()))
    torch.cuda.synchronize()
    t = A.double().cpu() @ W.double().cpu().t()
    yr = (t - t.mean(-1, keepdim=True)) / (t.std(-1, keepdim=True) + 1e-6)
    assert rel_err(y.cpu().numpy(), yr.numpy()) < 2e-5
