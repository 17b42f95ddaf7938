"""Launch a few mmnas_gemm shapes repeatedly (for `rocprofv3 --pmc ...` passes; tuning aid).
Shapes: NT 6400x2048x512, NT 6400x512x2048, NT 8192x2048x2048 with 64^2 and 128^2 tiles, TN 512x2048x6400."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import ops  # noqa: E402
import mmnas_amd._lib as L  # noqa: E402


def go(layout, M, N, K, tile, reps=10, acc=False):
    os.environ['MMNAS_GEMM_TILE'] = str(tile)
    L.lib().mmnas_gemm_reload_tuning()
    dev = 'cuda'
    if layout == 'NT':
        a, b, lda, ldb = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), K, K
    elif layout == 'NN':
        a, b, lda, ldb = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), K, N
    else:
        a, b, lda, ldb = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), M, N
    c = torch.zeros(M, N, device=dev)
    for _ in range(reps):
        ops.gemm({'NT': 0, 'NN': 1, 'TN': 2}[layout], [dict(M=M, A=[a], B=[b], C=c)], N, K, lda, ldb, N, accumulate=acc)
    torch.cuda.synchronize()


if __name__ == '__main__':
    if os.environ.get('GEMM_PMC_SWEEP'):   # fabric traffic vs tile order / XCD mapping on one shape
        for xcd in (1, 0):
            for gm in (1, 8, 16):
                os.environ['MMNAS_GEMM_XCD'] = str(xcd)
                os.environ['MMNAS_GEMM_GM'] = str(gm)
                go('NT', 6400, 2048, 512, 64, reps=4)
                go('NT', 6400, 512, 512, 64, reps=4)
        sys.exit(0)
    if os.environ.get('GEMM_PMC_LAYOUTS'):   # the three operand layouts on the supernet / training shapes (LDS counters)
        for d in (256, 512):
            go('NT', 6400, d, d, 64, reps=4)
            go('NT', 6400, 4 * d, d, 64, reps=4)
            go('NN', 6400, d, 4 * d, 64, reps=4)
            go('TN', d, 4 * d, 6400, 64, reps=4, acc=True)
        go('NT', 8192, 2048, 2048, 128, reps=4)
        sys.exit(0)
    for tile in (64, 128):
        go('NT', 8192, 2048, 2048, tile)
        go('NT', 6400, 2048, 512, tile)
        go('NT', 6400, 512, 2048, tile)
    go('TN', 512, 2048, 6400, 64, acc=True)
