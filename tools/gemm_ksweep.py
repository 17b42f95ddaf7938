"""Per-K-tile cost of the split-operand GEMM at 1 / 2 / 3 resident workgroups per CU (tuning aid).

N = 64 and M = 64 * 256 * w give exactly w workgroups of one 64x64 tile on each of the 256 CUs; sweeping K separates the
launch's fixed part (intercept) from the K loop (slope, per 32-deep K-tile).  NT layout, graph-replayed launches."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import ops  # noqa: E402
import mmnas_amd._lib as L  # noqa: E402


def timed(desc, iters=40):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        L.check(L.lib().mmnas_gemm(C.byref(desc), L.stream()))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                L.check(L.lib().mmnas_gemm(C.byref(desc), L.stream()))
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * iters)


if __name__ == '__main__':
    N = int(os.environ.get('KSWEEP_N', '64'))
    LAY = os.environ.get('KSWEEP_LAYOUT', 'NT')   # NT: both operands K-contiguous; NN: B as [K][N]; TN: both as [K][rows]
    for w in (1, 2, 3, 4, 6):
        M = 64 * 256 * w * 64 // N
        row = []
        for K in (64, 128, 256, 512, 1024, 2048):
            c = torch.zeros(M, N, device='cuda')
            if LAY == 'NT':
                a, b = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda')
                d = ops.gemm_desc(L.GEMM_NT, [dict(M=M, A=[a], B=[b], C=c)], N, K, K, K, N)
            elif LAY == 'NN':
                a, b = torch.randn(M, K, device='cuda'), torch.randn(K, N, device='cuda')
                d = ops.gemm_desc(L.GEMM_NN, [dict(M=M, A=[a], B=[b], C=c)], N, K, K, N, N)
            else:
                a, b = torch.randn(K, M, device='cuda'), torch.randn(K, N, device='cuda')
                d = ops.gemm_desc(L.GEMM_TN, [dict(M=M, A=[a], B=[b], C=c)], N, K, M, N, N)
            row.append((K, timed(d)))
        (k0, t0), (k1, t1) = row[2], row[-1]
        slope = (t1 - t0) / ((k1 - k0) / 32)
        print('%s N=%d  %d workgroups/CU (M=%d): ' % (LAY, N, w, M) + '  '.join('K=%d %.1f us' % kt for kt in row)
              + '  | per K-tile %.3f us = %.0f cycles at 2.4 GHz, intercept %.1f us' % (slope, slope * 2400, t0 - slope * k0 / 32), flush=True)
