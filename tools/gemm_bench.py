"""GPU microbenchmark of mmnas_gemm on the shapes the VQA workloads launch (tuning aid).
Prints one line per (shape, tile): TFLOP/s and microseconds, timed with events over many launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import ops  # noqa: E402
import mmnas_amd._lib as L  # noqa: E402

DEV = 'cuda'


def run(layout, Ms, N, K, nseg=1, split=1, iters=30):
    lay = {'NT': L.GEMM_NT, 'NN': L.GEMM_NN, 'TN': L.GEMM_TN}[layout]
    groups = []
    for M in Ms:
        if layout == 'NT':
            A = [torch.randn(M, K, device=DEV) for _ in range(nseg)]
            B = [torch.randn(N, K, device=DEV) for _ in range(nseg)]
            lda, ldb = K, K
        elif layout == 'NN':
            A = [torch.randn(M, K, device=DEV) for _ in range(nseg)]
            B = [torch.randn(K, N, device=DEV) for _ in range(nseg)]
            lda, ldb = K, N
        else:
            A = [torch.randn(K, M, device=DEV) for _ in range(nseg)]
            B = [torch.randn(K, N, device=DEV) for _ in range(nseg)]
            lda, ldb = M, N
        groups.append(dict(M=M, A=A, B=B, C=torch.zeros(M, N, device=DEV)))
    flops = 2.0 * sum(Ms) * N * K * nseg
    res = {}
    for tile in (64, 128):
        os.environ['MMNAS_GEMM_TILE'] = str(tile)
        L.lib().mmnas_gemm_reload_tuning()
        for _ in range(3):
            ops.gemm(lay, groups, N, K, lda, ldb, N, nseg=nseg, accumulate=(layout == 'TN'))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.gemm(lay, groups, N, K, lda, ldb, N, nseg=nseg, accumulate=(layout == 'TN'))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        res[tile] = (flops / us / 1e6, us)
    os.environ.pop('MMNAS_GEMM_TILE', None)
    L.lib().mmnas_gemm_reload_tuning()
    print('%-3s M=%-16s N=%-5d K=%-5d nseg=%d split=%-3d | t64: %6.1f TF %7.1f us | t128: %6.1f TF %7.1f us'
          % (layout, Ms, N, K, nseg, split, res[64][0], res[64][1], res[128][0], res[128][1]), flush=True)


if __name__ == '__main__':
    for d in (512, 256):
        print('--- d =', d)
        for M in (6400, 896):
            run('NT', [M, M, M], d, d)
            run('NT', [M], d, d)
            run('NT', [M], 4 * d, d)
            run('NT', [M], d, 4 * d)
            run('NN', [M], d, d)
            run('NN', [M], d, d, nseg=3)
            run('NN', [M], 4 * d, d)
            run('NN', [M], d, 4 * d)
            for sp in (1, 4, 8, 16, 32):
                run('TN', [d], d, M, split=sp)
            for sp in (1, 4, 8, 16):
                run('TN', [d, d, d], d, M, split=sp)
                run('TN', [4 * d], d, M, split=sp)
                run('TN', [d], 4 * d, M, split=sp)
        run('NT', [6400, 896, 896], d, d)
        run('NT', [6400], d, 2048)
        run('TN', [d], 2048, 6400, split=4)
    run('NT', [4096], 4096, 4096)
    run('NN', [4096], 4096, 4096)
    run('TN', [4096], 4096, 4096)
