"""Tuning aid: times mmnas_layernorm_fwd / _bwd on the workloads' row shapes (HIP events around 100 launches).

    python tools/ln_bench.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import _lib as L  # noqa: E402


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
    shapes = [(6400, 256, 0.1), (6400, 256, 0.0), (896, 256, 0.1), (6400, 512, 0.1), (896, 512, 0.1), (2304, 512, 0.1)]
    if len(sys.argv) == 4:   # one shape (under rocprofv3: the per-kernel averages then belong to it)
        shapes = [(int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]))]
    for (M, d, drop) in shapes:
        x, dy = torch.randn(M, d, device=dev), torch.randn(M, d, device=dev)
        a, b = torch.randn(d, device=dev), torch.randn(d, device=dev)
        y, dx, dd = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        da, db, dc = torch.zeros(d, device=dev), torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        ws = torch.empty(lib.mmnas_layernorm_bwd_ws_floats(M, d), device=dev)
        st = L.stream()
        t_f = timed(lambda: lib.mmnas_layernorm_fwd(L.fptr(x), L.fptr(a), L.fptr(b), L.fptr(y), M, d, 1e-6, st))
        t_b = timed(lambda: lib.mmnas_layernorm_bwd(L.fptr(x), L.fptr(a), L.fptr(dy), L.fptr(dx), L.fptr(da), L.fptr(db),
                                                    L.fptr(dd) if drop else None, L.fptr(dc) if drop else None, L.fptr(ws),
                                                    drop, 1234, 1, M, d, 1e-6, st))
        byt_f, byt_b = 8.0 * M * d, 4.0 * M * d * (4 if drop else 3)
        print('M=%5d d=%4d drop=%.1f | fwd %5.1f us %5.2f TB/s | bwd (+ reduce launch) %5.1f us %5.2f TB/s'
              % (M, d, drop, t_f, byt_f / t_f / 1e6, t_b, byt_b / t_b / 1e6))


if __name__ == '__main__':
    main()
