"""Accuracy and speed of the bf16-split GEMM modes (MMNAS_GEMM_SPLIT=3|6) next to the fp32-MFMA default.

    python tools/gemm_split.py            # error against an fp64 product + interleaved timing per mode

Error norm: max |c - ref| / max |ref| (the parity norm of SURVEY 8d).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import _lib as L, ops  # noqa: E402

MODES = [0, 3, 6]


def set_mode(m, extra=None):
    os.environ['MMNAS_GEMM_SPLIT'] = str(m)
    for k, v in (extra or {}).items():
        os.environ[k] = str(v)
    L.check(L.lib().mmnas_gemm_reload_tuning())


def run(layout, M, N, K, a, b, c, accumulate=False):
    if layout == 'NT':
        ops.gemm(L.GEMM_NT, [dict(M=M, A=[a], B=[b], C=c)], N, K, K, K, N)
    elif layout == 'NN':
        ops.gemm(L.GEMM_NN, [dict(M=M, A=[a], B=[b], C=c)], N, K, K, N, N)
    else:
        ops.gemm(L.GEMM_TN, [dict(M=M, A=[a], B=[b], C=c)], N, K, M, N, N, accumulate=accumulate)


def main():
    dev = 'cuda'
    torch.manual_seed(0)
    shapes = []
    for d in (512,):
        for M in (6400, 896):
            shapes += [('NT', M, d, d), ('NT', M, 4 * d, d), ('NT', M, d, 4 * d), ('NN', M, d, d), ('NN', M, 4 * d, d),
                       ('NN', M, d, 4 * d), ('TN', d, d, M), ('TN', 4 * d, d, M), ('TN', d, 4 * d, M)]
    shapes += [('NT', 6400, 512, 2048), ('NT', 200, 96, 64), ('NN', 333 * 4, 128, 96), ('TN', 132, 260, 800)]
    tiles = [int(x) for x in os.environ.get('TILES', '0').split(',')]
    print('%-4s %6s %6s %6s | %s' % ('lay', 'M', 'N', 'K', '  '.join('split%d t%-3d err      us    TF/s' % (m, t) for m in MODES for t in tiles)))
    for lay, M, N, K in shapes:
        if lay == 'NT':
            a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
            ref = a.double() @ b.double().t()
        elif lay == 'NN':
            a, b = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
            ref = a.double() @ b.double()
        else:
            a, b = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
            ref = a.double().t() @ b.double()
        scale = float(ref.abs().max())
        c = torch.zeros(M, N, device=dev)
        acc = lay == 'TN'
        cells = []
        cfgs = [(m, t) for m in MODES for t in tiles]
        ev = {}
        for m, t in cfgs:
            set_mode(m, {'MMNAS_GEMM_TILE': t})
            c.zero_()
            run(lay, M, N, K, a, b, c, acc)
            err = float((c.double() - ref).abs().max()) / scale
            for _ in range(5):
                run(lay, M, N, K, a, b, c, acc)
            ev[(m, t)] = [err, 0.0]
        reps = 10
        for _ in range(reps):          # interleaved: clock drift hits every mode alike
            for m, t in cfgs:
                set_mode(m, {'MMNAS_GEMM_TILE': t})
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(lay, M, N, K, a, b, c, acc)
                e1.record()
                e1.synchronize()
                ev[(m, t)][1] += e0.elapsed_time(e1) * 100.0 / reps   # us per launch
        for m, t in cfgs:
            err, us = ev[(m, t)]
            cells.append('%8.1e %7.1f %6.1f' % (err, us, 2.0 * M * N * K / us * 1e-6))
        print('%-4s %6d %6d %6d | %s' % (lay, M, N, K, '   '.join(cells)), flush=True)


if __name__ == '__main__':
    main()
