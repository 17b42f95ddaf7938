"""Fixed cost of a launch on this box, graph-replayed back to back (tuning aid): an elementwise torch kernel on one
element, the GEMM on one 64x64x32 tile, on 400 tiles of one K-tile, and the same with K = 64 / 256."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import ops  # noqa: E402
import mmnas_amd._lib as L  # noqa: E402


def timed(fn, iters=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * iters)


def gemm(M, N, K, **kw):
    a, b, c = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda'), torch.zeros(M, N, device='cuda')
    grp = dict(M=M, A=[a], B=[b], C=c)
    if kw.pop('bias', False):
        grp['bias'] = torch.randn(N, device='cuda')
    if kw.pop('residual', False):
        grp['residual'] = torch.randn(M, N, device='cuda')
        kw['ldres'] = N
    d = ops.gemm_desc(L.GEMM_NT, [grp], N, K, K, K, N, **kw)
    keep = (a, b, c, grp)
    return lambda: (L.check(L.lib().mmnas_gemm(C.byref(d), L.stream())), keep)[0]


if __name__ == '__main__':
    x = torch.zeros(1, device='cuda')
    print('torch add_ on one element          %6.2f us' % timed(lambda: x.add_(1.0)))
    big = torch.zeros(6400 * 256, device='cuda')
    print('torch add_ on 6400 x 256           %6.2f us' % timed(lambda: big.add_(1.0)))
    for M, N, K in ((64, 64, 32), (64, 64, 256), (6400, 256, 32), (6400, 256, 64), (6400, 256, 128), (6400, 256, 256)):
        print('NT %5d x %4d x %4d             %6.2f us' % (M, N, K, timed(gemm(M, N, K))))
    print('NT  6400 x  256 x  256 + bias + residual  %6.2f us' % timed(gemm(6400, 256, 256, bias=True, residual=True)))
