"""Tuning aid: the row-panel product with the LayerNorm inside (mmnas_gemm_ln, gemmln.hip) against the two launches it
replaces (mmnas_gemm + mmnas_layernorm_fwd), on the d = 256 shapes of the supernet step.  HIP events around 100 calls.

    python tools/gemm_ln_bench.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MMNAS_GEMM_LN_MINM', '0'); os.environ.setdefault('MMNAS_GEMM_LN_MAXK', '65536')   # (the small row counts run the panel kernel too, for comparison)
from mmnas_amd import _lib as L, ops  # noqa: E402


def timed(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    dev = 'cuda'
    lib = L.lib()
    shapes = [(6400, 256), (6400, 512), (6400, 1024), (896, 256), (896, 1024)]
    if len(sys.argv) == 3:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
    for (M, K) in shapes:
        N = 256
        A, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5
        b, R = torch.randn(N, device=dev), torch.randn(M, N, device=dev)
        la, lb = torch.randn(N, device=dev), torch.randn(N, device=dev)
        z, y = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        d = ops.gemm_desc(L.GEMM_NT, [dict(M=M, A=[A], B=[W], C=z, bias=b, residual=R)], N, K, K, K, N, drop=(0.1, 1234, 1), ldres=N)
        st = L.stream()
        res = {}
        for panel in (0, 1):
            lib.mmnas_set_gemm_ln(panel)
            res[panel] = timed(lambda: lib.mmnas_gemm_ln(C.byref(d), L.fptr(la), L.fptr(lb), L.fptr(y), 1e-6, st))
        lib.mmnas_set_gemm_ln(1)
        Wp = ops.split_planes(W)
        dp = ops.gemm_desc(L.GEMM_NT, [dict(M=M, A=[A], B=[Wp], C=z, bias=b, residual=R)], N, K, K, K, N, drop=(0.1, 1234, 1), ldres=N,
                           b_planes=True)
        res[2] = timed(lambda: lib.mmnas_gemm_ln(C.byref(dp), L.fptr(la), L.fptr(lb), L.fptr(y), 1e-6, st))
        fl = 2.0 * M * N * K
        print('M=%5d K=%5d | gemm + layernorm %6.1f us (%5.1f TF/s) | panel kernel %6.1f us (%5.1f TF/s) %+5.1f us | panel, W as planes %6.1f us (%5.1f TF/s) %+5.1f us'
              % (M, K, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6, res[1] - res[0], res[2], fl / res[2] / 1e6, res[2] - res[0]))


if __name__ == '__main__':
    main()
