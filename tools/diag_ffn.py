"""GPU diagnostic: locate the full-size FFN dx discrepancy (where, which tile) and check big-K GEMMs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden import cases
from tests import oracle_runner as R
from tests.test_ops_gpu import run_hip_op
from mmnas_amd import ops
import mmnas_amd._lib as L

def where(a, b, name):
    e = np.abs(a.astype(np.float64) - b.astype(np.float64))
    idx = np.unravel_index(np.argmax(e), e.shape)
    print(name, 'max abs err %.3e at %s (got %.6f ref %.6f), ref max %.3f, n(e>1e-3*max)=%d' % (
        e.max(), idx, a[idx], b[idx], np.abs(b).max(), int((e > 1e-3 * np.abs(b).max()).sum())))
    return e

case = cases.op_case('feed_forward', True, True, 777, dict(B=64, Sx=100, Sy=14, HSIZE=512))
got = run_hip_op(case)
ref = R.run_oracle_op(case)
ref64 = R.run_oracle_op(case, dtype=torch.float64)
for k in ('out', 'dx'):
    where(got[k], ref[k], k + ' hip-vs-oracle32')
    where(got[k], ref64[k], k + ' hip-vs-oracle64')
    where(ref[k], ref64[k], k + ' oracle32-vs-oracle64')
e = where(got['dx'], ref64['dx'], 'dx')
e2 = e.reshape(6400, 512)
rows = np.where(e2.max(1) > 1e-3 * np.abs(ref64['dx']).max())[0]
print('bad rows', rows[:40], 'count', len(rows))
cols = np.where(e2.max(0) > 1e-3 * np.abs(ref64['dx']).max())[0]
print('bad cols', cols[:40], 'count', len(cols))

rs = np.random.RandomState(0)
for (M, N, K) in ((6400, 512, 2048), (6400, 2048, 512)):
    for lay, name in ((L.GEMM_NT, 'NT'), (L.GEMM_NN, 'NN'), (L.GEMM_TN, 'TN')):
        if name == 'NT': A, B = rs.standard_normal((M, K)), rs.standard_normal((N, K)); refc = A @ B.T; lda, ldb = K, K
        elif name == 'NN': A, B = rs.standard_normal((M, K)), rs.standard_normal((K, N)); refc = A @ B; lda, ldb = K, N
        else: A, B = rs.standard_normal((K, M)), rs.standard_normal((K, N)); refc = A.T @ B; lda, ldb = M, N
        Ad = torch.from_numpy(A.astype(np.float32)).cuda(); Bd = torch.from_numpy(B.astype(np.float32)).cuda()
        refc = Ad.double().cpu().numpy() @ Bd.double().cpu().numpy().T if name == 'NT' else (
            Ad.double().cpu().numpy() @ Bd.double().cpu().numpy() if name == 'NN' else Ad.double().cpu().numpy().T @ Bd.double().cpu().numpy())
        C = torch.zeros(M, N, device='cuda')
        ops.gemm(lay, [dict(M=M, A=[Ad], B=[Bd], C=C)], N, K, lda, ldb, N, split_k=(4 if name == 'TN' else 1))
        where(C.cpu().numpy(), refc, 'gemm %s %dx%dx%d' % (name, M, N, K))
