"""Microbenchmark: mmnas_gemm on shapes that tile the 256 CUs exactly (upper bound for a balanced schedule)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import run  # noqa: E402

if __name__ == '__main__':
    for M in (8192, 6400):
        for (N, K) in ((512, 512), (512, 2048), (2048, 512), (2048, 2048)):
            run('NT', [M], N, K)
            run('NN', [M], N, K)
    for sp in (1, 2, 4):
        run('TN', [512], 2048, 8192, split=sp)
        run('TN', [2048], 2048, 8192, split=sp)
